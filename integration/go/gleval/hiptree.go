package gleval

// hiptree.go -- pure Go (no cgo, no build tag): the flattened CSG tree that crosses the C ABI of libgsdfhip.so, and the
// walker the node packages (gsdf, forge/threads) append themselves to. Goes into github.com/soypat/gsdf/gleval.
//
// HIPNode is include/gsdf_program.h's gsdf_node field for field (48 bytes; gpu_hip.go asserts the size against the C
// struct and hands &Nodes[0] to the library as *C.gsdf_node). HIPOp is enum gsdf_op; its numbering is part of the ABI
// (the header pins GSDF_SPHERE == 1, GSDF_UNION == 7, GSDF_SCREW == 26, GSDF_POLY2D == 38, GSDF_OP_COUNT == 54 with
// static asserts; tests/test_integration_lock.py checks every constant below against the header).
//
// Why not GLSL text as for SDF3Compute (gsdfaux/gsdfaux.go:122-126): the shader source prints parameters with nine
// decimals (glbuild/glbuild.go:937-956); the node structs' own float32 fields are exact.

import (
	"errors"
	"reflect"

	"github.com/soypat/gsdf/glbuild"
)

// HIPOp is enum gsdf_op of include/gsdf_program.h.
type HIPOp uint16

const (
	HIPOpInvalid HIPOp = iota
	// 3D primitives (cpu_evaluators.go:20-105)
	HIPSphere   // p0=r
	HIPBox      // p0..2=dims p3=round
	HIPBoxFrame // p0..2=dims p3=e
	HIPTorus    // p0=rGreater p1=rLesser
	HIPCylinder // p0=r p1=h p2=round
	HIPHex      // p0=side p1=h
	// 3D booleans (cpu_evaluators.go:124-286)
	HIPUnion
	HIPIntersect
	HIPDiff
	HIPXor
	HIPSmoothUnion     // p0=k
	HIPSmoothDiff      // p0=k
	HIPSmoothIntersect // p0=k
	// 3D unary operations (cpu_evaluators.go:288-504,1042,1257)
	HIPScale     // p0=scale
	HIPSymmetry  // p0=bit mask (1=x 2=y 4=z)
	HIPArray     // p0..2=d p3..5=nx,ny,nz
	HIPElongate  // p0..2=h
	HIPShell     // p0=thick
	HIPOffset    // p0=off
	HIPTranslate // p0..2=p
	HIPTransform // aux[0..15]=tInv, row major
	HIPCircArray // p0=nInst p1=circleDiv
	HIPTwist     // p0=k
	// 2D -> 3D (cpu_evaluators.go:506-549, forge/threads/threads.go:141-181)
	HIPExtrusion  // p0=h
	HIPRevolution // p0=off
	HIPScrew      // p0=pitch p1=lead p2=lengthDiv2 p3=taper
	// 2D primitives (cpu_evaluators.go:551-818,1145)
	HIPLine2D       // p0,1=a p2,3=b p4=width
	HIPArc2D        // p0=radius p1=angle p2=thick
	HIPQuadBezier2D // p0,1=a p2,3=b p4,5=c p6=thick
	HIPCircle2D     // p0=r
	HIPEqTri2D      // p0=hTri
	HIPRect2D       // p0,1=d
	HIPDiamond2D    // p0,1=d
	HIPX2D          // p0=dim p1=thick
	HIPHex2D        // p0=side
	HIPOct2D        // p0=c
	HIPEllipse2D    // p0=a p1=b
	HIPPoly2D       // aux = x0,y0,x1,y1,...
	HIPLines2D      // p0=width; aux = ax,ay,bx,by per segment
	// 2D operations (cpu_evaluators.go:821-1255)
	HIPUnion2D
	HIPIntersect2D
	HIPDiff2D
	HIPXor2D
	HIPArray2D          // p0,1=d p2,3=nx,ny
	HIPOffset2D         // p0=f
	HIPTranslate2D      // p0,1=p
	HIPSymmetry2D       // p0=bit mask (1=x 2=y)
	HIPAnnulus2D        // p0=r
	HIPCircArray2D      // p0=nInst p1=circleDiv
	HIPTranslateMulti2D // aux = dx,dy per displacement
	HIPRotation2D       // p0..3 = tInv x00,x01,x10,x11
	HIPScale2D          // p0=scale
	HIPElongate2D       // p0,1=h
	HIPOpCount
)

// HIPNodeNParam is GSDF_NODE_NPARAM.
const HIPNodeNParam = 8

// HIPNode is gsdf_node: one per reference node, the node struct's own fields verbatim.
type HIPNode struct {
	Op      HIPOp  // enum gsdf_op
	NChild  uint16 // number of children
	LinkOff uint32 // first child slot in Links
	AuxOff  uint32 // first float in Aux
	AuxLen  uint32 // number of floats in Aux
	P       [HIPNodeNParam]float32
}

// HIPTree is gsdf_tree without the borrowed pointers: what NewHIPSDF3 / NewHIPSDF2 hand to gsdf_hip_program_create.
type HIPTree struct {
	Nodes []HIPNode
	Links []uint32
	Aux   []float32
	Root  uint32
}

// HIPNodeAppender is implemented by every node type of package gsdf and by forge/threads' screw
// (hip_flatten.go in those packages): append yourself -- children first, through the flattener -- and
// return your index in Tree.Nodes.
type HIPNodeAppender interface {
	AppendHIPNodes(f *HIPFlattener) (uint32, error)
}

// HIPFlattener walks a shader DAG into a HIPTree. Shared sub-shaders (the same pointer reached twice) become one
// node: DAGs stay DAGs.
type HIPFlattener struct {
	Tree HIPTree
	memo map[any]uint32
}

var errNotHIPNode = errors.New("gsdf_hip: shader has no AppendHIPNodes (not a gsdf / forge/threads node)")

func (f *HIPFlattener) shader(s glbuild.Shader) (uint32, error) {
	// glbuild's decorators (nameOverloadShader3D/2D, CachedShader3D/2D, overloadBounds3/2: glbuild/glbuild.go:1087-1333)
	// forward evaluation to the node they wrap; the wrapped node is what is flattened (glbuild.UnwrapHIP is
	// unwraproot, glbuild/glbuild.go:1366-1382, exported by glbuild/hip_unwrap.go).
	s = glbuild.UnwrapHIP(s)
	if s == nil {
		return 0, errors.New("gsdf_hip: nil shader")
	}
	key := reflect.ValueOf(s).Kind() == reflect.Pointer
	if key {
		if id, ok := f.memo[s]; ok {
			return id, nil
		}
	}
	n, ok := s.(HIPNodeAppender)
	if !ok {
		return 0, errNotHIPNode
	}
	id, err := n.AppendHIPNodes(f)
	if err != nil {
		return 0, err
	}
	if key {
		if f.memo == nil {
			f.memo = make(map[any]uint32)
		}
		f.memo[s] = id
	}
	return id, nil
}

// Shader3D flattens a 3D child and returns its node index.
func (f *HIPFlattener) Shader3D(s glbuild.Shader3D) (uint32, error) { return f.shader(s) }

// Shader2D flattens a 2D child and returns its node index.
func (f *HIPFlattener) Shader2D(s glbuild.Shader2D) (uint32, error) { return f.shader(s) }

func (f *HIPFlattener) push(op HIPOp, children []uint32, aux []float32, p []float32) (uint32, error) {
	if len(p) > HIPNodeNParam {
		return 0, errors.New("gsdf_hip: too many node parameters")
	}
	n := HIPNode{Op: op, NChild: uint16(len(children)), LinkOff: uint32(len(f.Tree.Links)), AuxOff: uint32(len(f.Tree.Aux)), AuxLen: uint32(len(aux))}
	copy(n.P[:], p)
	f.Tree.Links = append(f.Tree.Links, children...)
	f.Tree.Aux = append(f.Tree.Aux, aux...)
	f.Tree.Nodes = append(f.Tree.Nodes, n)
	return uint32(len(f.Tree.Nodes) - 1), nil
}

// Leaf appends a primitive: parameters only.
func (f *HIPFlattener) Leaf(op HIPOp, p ...float32) (uint32, error) { return f.push(op, nil, nil, p) }

// LeafAux appends a primitive with a variable-length float payload (polygon vertices, segments).
func (f *HIPFlattener) LeafAux(op HIPOp, aux []float32, p ...float32) (uint32, error) {
	return f.push(op, nil, aux, p)
}

// Op3 appends an operation over 3D children (in the order its Evaluate visits them).
func (f *HIPFlattener) Op3(op HIPOp, children []glbuild.Shader3D, aux []float32, p ...float32) (uint32, error) {
	ids := make([]uint32, len(children))
	for i, c := range children {
		id, err := f.Shader3D(c)
		if err != nil {
			return 0, err
		}
		ids[i] = id
	}
	return f.push(op, ids, aux, p)
}

// Op2 appends an operation over 2D children (2D operations, and extrusion / revolution / screw).
func (f *HIPFlattener) Op2(op HIPOp, children []glbuild.Shader2D, aux []float32, p ...float32) (uint32, error) {
	ids := make([]uint32, len(children))
	for i, c := range children {
		id, err := f.Shader2D(c)
		if err != nil {
			return 0, err
		}
		ids[i] = id
	}
	return f.push(op, ids, aux, p)
}

// FlattenHIP3 flattens a 3D shader tree; FlattenHIP2 a 2D one. The root is the last node appended.
func FlattenHIP3(s glbuild.Shader3D) (HIPTree, error) {
	var f HIPFlattener
	root, err := f.Shader3D(s)
	if err != nil {
		return HIPTree{}, err
	}
	f.Tree.Root = root
	return f.Tree, nil
}

func FlattenHIP2(s glbuild.Shader2D) (HIPTree, error) {
	var f HIPFlattener
	root, err := f.Shader2D(s)
	if err != nil {
		return HIPTree{}, err
	}
	f.Tree.Root = root
	return f.Tree, nil
}
