package gsdf

// hip_flatten.go -- goes into github.com/soypat/gsdf (package gsdf). Pure Go, no build tag: one AppendHIPNodes per node
// type of this package -- a field copy into gleval.HIPNode (= gsdf_node, include/gsdf_program.h) -- found through
// gleval.HIPFlattener, which recurses into children and unwraps glbuild's decorators. Node parameters are unexported
// struct fields, which is why these methods must live here (and forge/threads/hip_flatten.go for `screw`).
//
// What is copied is what the CPU evaluators read (cpu_evaluators.go); anything they derive per call (cylinder args(),
// 1/scale, polygon edge constants, ...) the library derives the same way from the same fields. Field <-> p[] mapping:
// the table in include/gsdf_program.h. tests/test_integration_lock.py checks, against the reference's struct
// declarations, that every node type below exists, that every field named here is a field of it, and that no node type
// of primitives*.go / operations*.go is missing.

import (
	"github.com/soypat/geometry/ms2"
	"github.com/soypat/gsdf/glbuild"
	"github.com/soypat/gsdf/gleval"
)

// FlattenHIP flattens a 3D shader tree for gleval.NewHIPSDF3 (replaces Programmer.WriteComputeSDF3 +
// NewComputeGPUSDF3's GLSL compile, gsdfaux/gsdfaux.go:122-126).
func FlattenHIP(s glbuild.Shader3D) (gleval.HIPTree, error) { return gleval.FlattenHIP3(s) }

// FlattenHIP2D is the 2D counterpart, for gleval.NewHIPSDF2.
func FlattenHIP2D(s glbuild.Shader2D) (gleval.HIPTree, error) { return gleval.FlattenHIP2(s) }

func hipVec2s(v []ms2.Vec) []float32 {
	out := make([]float32, 0, 2*len(v))
	for _, p := range v {
		out = append(out, p.X, p.Y)
	}
	return out
}

// ---- 3D primitives (primitives.go:23-300; cpu_evaluators.go:20-105) ----

func (s *sphere) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPSphere, s.r)
}

func (s *box) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPBox, s.dims.X, s.dims.Y, s.dims.Z, s.round)
}

func (s *boxframe) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPBoxFrame, s.dims.X, s.dims.Y, s.dims.Z, s.e) // e as stored: NewBoxFrame halved it (primitives.go:255)
}

func (s *torus) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPTorus, s.rGreater, s.rLesser)
}

func (s *cylinder) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPCylinder, s.r, s.h, s.round) // args() is derived by the library (primitives.go:139-145)
}

func (s *hex) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPHex, s.side, s.h)
}

// ---- 3D booleans (operations.go:27-681; cpu_evaluators.go:124-286) ----

func (s *OpUnion) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPUnion, s.joined, nil)
}

func (s *intersect) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPIntersect, []glbuild.Shader3D{s.s1, s.s2}, nil)
}

func (s *diff) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPDiff, []glbuild.Shader3D{s.s1, s.s2}, nil)
}

func (s *xor) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPXor, []glbuild.Shader3D{s.s1, s.s2}, nil)
}

func (s *smoothUnion) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPSmoothUnion, []glbuild.Shader3D{s.s1, s.s2}, nil, s.k)
}

// smoothDiff embeds diff and smoothIntersect embeds intersect (operations.go:618,650): their own methods shadow the
// promoted ones.
func (s *smoothDiff) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPSmoothDiff, []glbuild.Shader3D{s.s1, s.s2}, nil, s.k)
}

func (s *smoothIntersect) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPSmoothIntersect, []glbuild.Shader3D{s.s1, s.s2}, nil, s.k)
}

// ---- 3D unary operations (operations.go:252-880; cpu_evaluators.go:288-504,1042-1092,1257-1274) ----

func (s *scale) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPScale, []glbuild.Shader3D{s.s}, nil, s.scale)
}

func (s *symmetry) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPSymmetry, []glbuild.Shader3D{s.s}, nil, float32(s.xyz)) // XYZBits: 1=X 2=Y 4=Z (glbuild.go:1031-1037)
}

func (s *array) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPArray, []glbuild.Shader3D{s.s}, nil, s.d.X, s.d.Y, s.d.Z, float32(s.nx), float32(s.ny), float32(s.nz))
}

func (s *elongate) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPElongate, []glbuild.Shader3D{s.s}, nil, s.h.X, s.h.Y, s.h.Z)
}

func (s *shell) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPShell, []glbuild.Shader3D{s.s}, nil, s.thick)
}

func (s *offset) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPOffset, []glbuild.Shader3D{s.s}, nil, s.off)
}

func (s *translate) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPTranslate, []glbuild.Shader3D{s.s}, nil, s.p.X, s.p.Y, s.p.Z)
}

func (s *transform) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	a := s.tInv.Array() // row major x00..x33: the inverse the CPU evaluator multiplies by (cpu_evaluators.go:495-497)
	return f.Op3(gleval.HIPTransform, []glbuild.Shader3D{s.s}, a[:])
}

func (s *circarray) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPCircArray, []glbuild.Shader3D{s.s}, nil, float32(s.nInst), float32(s.circleDiv))
}

func (s *twist) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op3(gleval.HIPTwist, []glbuild.Shader3D{s.s}, nil, s.k)
}

// ---- 2D -> 3D (operations2d.go:114-207; cpu_evaluators.go:506-549) ----

func (s *extrusion) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPExtrusion, []glbuild.Shader2D{s.s}, nil, s.h)
}

func (s *revolution) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPRevolution, []glbuild.Shader2D{s.s2d}, nil, s.off)
}

// ---- 2D primitives (primitives2d.go:33-680; cpu_evaluators.go:551-818,1145-1160) ----

func (s *line2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPLine2D, s.a.X, s.a.Y, s.b.X, s.b.Y, s.width)
}

func (s *arc2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPArc2D, s.radius, s.angle, s.thick)
}

func (s *quadbezier2d) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPQuadBezier2D, s.a.X, s.a.Y, s.b.X, s.b.Y, s.c.X, s.c.Y, s.thick)
}

func (s *circle2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPCircle2D, s.r)
}

func (s *equilateralTri2d) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPEqTri2D, s.hTri)
}

func (s *rect2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPRect2D, s.d.X, s.d.Y)
}

func (s *diamond) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPDiamond2D, s.d.X, s.d.Y)
}

func (s *x2d) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPX2D, s.dim, s.thick)
}

func (s *hex2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPHex2D, s.side)
}

func (s *oct2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPOct2D, s.c)
}

func (s *ellipse2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Leaf(gleval.HIPEllipse2D, s.a, s.b)
}

// poly2D; polySSBO embeds it (primitives2d.go:536-539) and is evaluated by the same method (cpu_evaluators.go:793-818):
// the promoted AppendHIPNodes covers it.
func (s *poly2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.LeafAux(gleval.HIPPoly2D, hipVec2s(s.vert)) // x0,y0,x1,y1,... (the closing duplicate was dropped by NewPolygon, primitives2d.go:471-476)
}

// lines2D; lines2Dssbo embeds it (primitives2d.go:146-149): promoted likewise.
func (s *lines2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	aux := make([]float32, 0, 4*len(s.points))
	for _, seg := range s.points {
		aux = append(aux, seg[0].X, seg[0].Y, seg[1].X, seg[1].Y)
	}
	return f.LeafAux(gleval.HIPLines2D, aux, s.width)
}

// ---- 2D operations (operations2d.go:15-860; cpu_evaluators.go:821-1255) ----

func (s *OpUnion2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPUnion2D, s.joined, nil)
}

func (s *intersect2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPIntersect2D, []glbuild.Shader2D{s.s1, s.s2}, nil)
}

func (s *diff2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPDiff2D, []glbuild.Shader2D{s.s1, s.s2}, nil)
}

func (s *xor2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPXor2D, []glbuild.Shader2D{s.s1, s.s2}, nil)
}

func (s *array2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPArray2D, []glbuild.Shader2D{s.s}, nil, s.d.X, s.d.Y, float32(s.nx), float32(s.ny))
}

func (s *offset2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPOffset2D, []glbuild.Shader2D{s.s}, nil, s.f)
}

func (s *translate2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPTranslate2D, []glbuild.Shader2D{s.s}, nil, s.p.X, s.p.Y)
}

func (s *symmetry2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPSymmetry2D, []glbuild.Shader2D{s.s}, nil, float32(s.xy)) // XYZBits: 1=X 2=Y
}

func (s *annulus2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPAnnulus2D, []glbuild.Shader2D{s.s}, nil, s.r)
}

func (s *circarray2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPCircArray2D, []glbuild.Shader2D{s.s}, nil, float32(s.nInst), float32(s.circleDiv))
}

func (s *translateMulti2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPTranslateMulti2D, []glbuild.Shader2D{s.s}, hipVec2s(s.displacements))
}

func (s *rotation2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	a := s.tInv.Array() // x00,x01,x10,x11: the matrix the CPU evaluator multiplies by (cpu_evaluators.go:1186-1203)
	return f.Op2(gleval.HIPRotation2D, []glbuild.Shader2D{s.s}, nil, a[0], a[1], a[2], a[3])
}

func (s *scale2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPScale2D, []glbuild.Shader2D{s.s}, nil, s.scale)
}

func (s *elongate2D) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPElongate2D, []glbuild.Shader2D{s.s}, nil, s.h.X, s.h.Y)
}
