//go:build cgo && hip

package gsdfaux

// hip.go -- goes into github.com/soypat/gsdf/gsdfaux; built with `-tags hip`. renderShader3DHIP is RenderShader3D's
// (gsdfaux.go:62-237) evaluation + meshing + STL part on the HIP backend. The switch itself is three lines in
// gsdfaux.go, next to `if cfg.UseGPU {` (gsdfaux.go:93):
//
//	type RenderConfig struct { ...; UseHIP bool }                          // gsdfaux.go:25-39
//	if cfg.UseHIP { return renderShader3DHIP(s, cfg, log, fromStart) }     // before gsdfaux.go:93
//
// and a no-tag twin of this file (`//go:build !cgo || !hip`) whose renderShader3DHIP returns gleval.InitHIP's error.
// cfg.VisualOutput keeps the reference's code path (gsdfaux.go:176-204: GLSL text for ShaderToy, no evaluation).

import (
	"fmt"
	"os"
	"time"

	"github.com/soypat/gsdf"
	"github.com/soypat/gsdf/glbuild"
	"github.com/soypat/gsdf/gleval"
	"github.com/soypat/gsdf/glrender"
)

func renderShader3DHIP(s glbuild.Shader3D, cfg RenderConfig, log func(elapsed time.Duration, args ...any), fromStart func() time.Duration) error {
	log(0, "using HIP (MI355X)")
	terminate, err := gleval.InitHIP(0) // instead of Init1x1GLFW (gsdfaux.go:96): no window, no OS thread affinity
	if err != nil {
		return err
	}
	defer terminate()
	watch := stopwatch()
	tree, err := gsdf.FlattenHIP(s) // instead of Programmer.WriteComputeSDF3 (gsdfaux.go:122)
	if err != nil {
		return err
	}
	sdf, err := gleval.NewHIPSDF3(tree, s.Bounds(), gleval.HIPConfig{Specialize: true}) // instead of NewComputeGPUSDF3 (gsdfaux.go:126)
	if err != nil {
		return err
	}
	defer sdf.Close()
	log(watch(), "tree flattened and lowered,", len(tree.Nodes), "nodes")
	if cfg.STLOutput == nil {
		return nil
	}
	watch = stopwatch()
	var mesh *glrender.MeshHIP
	if cfg.renderer == renderWithFlatMC {
		mesh, err = glrender.NewFlatRendererHIP(sdf, cfg.Resolution, 4096, 1) // gsdfaux.go:160-168
	} else {
		mesh, err = glrender.NewOctreeRendererHIP(sdf, cfg.Resolution) // gsdfaux.go:170
	}
	if err != nil {
		return err
	}
	defer mesh.Close()
	if cfg.renderer == renderWithFlatMC {
		log(watch(), "evaluated SDF", mesh.Evaluations(), "times and rendered", mesh.NumTriangles(), "triangles with resolution", cfg.Resolution)
	} else {
		omitted := 8 * mesh.TotalPruned() // gsdfaux.go:220-223
		log(watch(), "evaluated SDF", mesh.Evaluations(), "times and rendered", mesh.NumTriangles(), "triangles with", percentUint64(omitted, mesh.Evaluations()+omitted), "percent evaluations omitted in octree pruning step with resolution", cfg.Resolution)
	}
	watch = stopwatch()
	if _, err = mesh.WriteBinarySTL(cfg.STLOutput); err != nil { // instead of RenderAll + WriteBinarySTL (gsdfaux.go:213,227)
		return fmt.Errorf("writing STL file: %s", err)
	}
	filename := "STL"
	if fp, ok := cfg.STLOutput.(*os.File); ok {
		filename = fp.Name()
	}
	log(watch(), "wrote", filename)
	log(fromStart(), "render done")
	return nil
}
