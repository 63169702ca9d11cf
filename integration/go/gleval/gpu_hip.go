//go:build cgo && hip

package gleval

// gpu_hip.go -- goes into github.com/soypat/gsdf/gleval; built with `-tags hip`. SDF3HIP / SDF2HIP: the drop-in for
// SDF3Compute / SDF2Compute (gpu.go:56-166) over libgsdfhip.so (include/gsdf_hip.h). Same contract as
// (*SDF3Compute).Evaluate (gpu.go:82-103): len(pos) != len(dist) -> errMismatchBufferLength, empty -> errEmptyBuffers
// (the library reports both with codes of their own, mapped back to THIS package's error values), dist fully
// overwritten, pos untouched, userData ignored, Evaluations() counts points evaluated.

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/gsdf_amd/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/gsdf_amd/gsdf_amd/csrc -lgsdfhip
// (this repository checked out under third_party/gsdf_amd: headers in include/, library in gsdf_amd/csrc/)
#include <stdlib.h>
#include "gsdf_hip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"

	"github.com/soypat/geometry/ms2"
	"github.com/soypat/geometry/ms3"
)

// HIPNode must be gsdf_node byte for byte: &Nodes[0] is handed to the library as *C.gsdf_node.
var _ [unsafe.Sizeof(C.gsdf_node{}) - unsafe.Sizeof(HIPNode{})]struct{}
var _ [unsafe.Sizeof(HIPNode{}) - unsafe.Sizeof(C.gsdf_node{})]struct{}
var _ [uint(C.GSDF_OP_COUNT) - uint(HIPOpCount)]struct{}
var _ [uint(HIPOpCount) - uint(C.GSDF_OP_COUNT)]struct{}

// InitHIP replaces Init1x1GLFW (gpu.go:21-32): no window, no OS-thread affinity (runtime.LockOSThread is not needed).
func InitHIP(device int) (terminate func(), err error) {
	if rc := C.gsdf_hip_init(C.int(device)); rc != 0 {
		return nil, hipErr(rc)
	}
	return func() {}, nil
}

func hipErr(rc C.int) error {
	switch rc {
	case C.GSDF_ERR_EMPTY_BUFFERS:
		return errEmptyBuffers // gleval.go:47
	case C.GSDF_ERR_LENGTH_MISMATCH:
		return errMismatchBufferLength // gleval.go:48
	}
	return errors.New("gsdf_hip: " + C.GoString(C.gsdf_hip_last_error()))
}

// HIPConfig stands where ComputeConfig does (gpu.go:28-33). Specialize asks for the per-tree kernel build -- the step
// NewComputeGPUSDF3 spends compiling GLSL (gpu.go:35-54) -- in the background; Wait blocks until it has landed.
type HIPConfig struct{ Specialize, Wait bool }

func newHIPProgram(t HIPTree, bb [6]float32, cfg HIPConfig) (*C.gsdf_program, error) {
	if len(t.Nodes) == 0 {
		return nil, errors.New("gsdf_hip: empty tree")
	}
	var ct C.gsdf_tree
	var pin runtime.Pinner // Go pointers stored inside a C struct must be pinned for the duration of the call
	defer pin.Unpin()
	pin.Pin(&t.Nodes[0])
	ct.nodes, ct.n_nodes = (*C.gsdf_node)(unsafe.Pointer(&t.Nodes[0])), C.uint32_t(len(t.Nodes))
	if len(t.Links) > 0 {
		pin.Pin(&t.Links[0])
		ct.links, ct.n_links = (*C.uint32_t)(unsafe.Pointer(&t.Links[0])), C.uint32_t(len(t.Links))
	}
	if len(t.Aux) > 0 {
		pin.Pin(&t.Aux[0])
		ct.aux, ct.n_aux = (*C.float)(unsafe.Pointer(&t.Aux[0])), C.uint32_t(len(t.Aux))
	}
	ct.root = C.uint32_t(t.Root)
	for i, v := range bb {
		ct.bb[i] = C.float(v)
	}
	var h *C.gsdf_program
	if rc := C.gsdf_hip_program_create(&ct, &h); rc != 0 { // the library copies / lowers the tree and retains nothing of it
		return nil, hipErr(rc)
	}
	// 1.6-4.3 s with the installed hipcc, a file read once GSDF_HIP_CACHE_DIR holds the tree's code object. The
	// examples mesh ONE tree once (examples/npt-flange/flange.go:61-98): the build runs in the background and the
	// handle evaluates through the interpreter kernels until it lands -- bit-identical results either way. Optional:
	// on failure the handle keeps the interpreter kernels.
	if cfg.Specialize {
		_ = C.gsdf_hip_program_specialize_async(h)
		if cfg.Wait {
			_ = C.gsdf_hip_program_specialize_poll(h, 1)
		}
	}
	return h, nil
}

// SDF3HIP implements SDF3 (gleval.go:15-24) and Evaluations() (type-asserted at gsdfaux/gsdfaux.go:219).
type SDF3HIP struct {
	h  *C.gsdf_program
	bb ms3.Box
}

// NewHIPSDF3 stands where NewComputeGPUSDF3 does (gpu.go:35-54): tree from gsdf.FlattenHIP instead of GLSL source.
func NewHIPSDF3(t HIPTree, bb ms3.Box, cfg HIPConfig) (*SDF3HIP, error) {
	h, err := newHIPProgram(t, [6]float32{bb.Min.X, bb.Min.Y, bb.Min.Z, bb.Max.X, bb.Max.Y, bb.Max.Z}, cfg)
	if err != nil {
		return nil, err
	}
	s := &SDF3HIP{h: h, bb: bb}
	runtime.SetFinalizer(s, func(s *SDF3HIP) { s.Close() })
	return s, nil
}

func (s *SDF3HIP) Bounds() ms3.Box     { return s.bb }
func (s *SDF3HIP) Evaluations() uint64 { return uint64(C.gsdf_hip_evaluations(s.h)) }

// Handle is the library's gsdf_program*, for the renderers of package glrender (cgo types do not cross packages).
func (s *SDF3HIP) Handle() unsafe.Pointer { return unsafe.Pointer(s.h) }

// Close frees the device program; the finalizer calls it too.
func (s *SDF3HIP) Close() {
	if s.h != nil {
		C.gsdf_hip_program_destroy(s.h)
		s.h = nil
	}
}

// Evaluate: the stride is unsafe.Sizeof(ms3.Vec{}) on purpose -- the reference declares the GLSL block as std140
// vec3[] (16-byte stride, glbuild/glbuild.go:195-197) but uploads []ms3.Vec verbatim (gpu_cgo.go:238); the C ABI
// makes the stride explicit so either layout works.
func (s *SDF3HIP) Evaluate(pos []ms3.Vec, dist []float32, userData any) error {
	var pp, dp unsafe.Pointer
	if len(pos) > 0 {
		pp = unsafe.Pointer(&pos[0])
	}
	if len(dist) > 0 {
		dp = unsafe.Pointer(&dist[0])
	}
	rc := C.gsdf_hip_eval3(s.h, pp, C.size_t(unsafe.Sizeof(ms3.Vec{})), C.size_t(len(pos)), (*C.float)(dp), C.size_t(len(dist)))
	if rc != 0 {
		return hipErr(rc)
	}
	return nil
}

// SDF2HIP implements SDF2 (gleval.go:26-37): the drop-in for SDF2Compute (gpu.go:105-166).
type SDF2HIP struct {
	h  *C.gsdf_program
	bb ms2.Box
}

func NewHIPSDF2(t HIPTree, bb ms2.Box, cfg HIPConfig) (*SDF2HIP, error) {
	h, err := newHIPProgram(t, [6]float32{bb.Min.X, bb.Min.Y, 0, bb.Max.X, bb.Max.Y, 0}, cfg)
	if err != nil {
		return nil, err
	}
	s := &SDF2HIP{h: h, bb: bb}
	runtime.SetFinalizer(s, func(s *SDF2HIP) { s.Close() })
	return s, nil
}

func (s *SDF2HIP) Bounds() ms2.Box        { return s.bb }
func (s *SDF2HIP) Evaluations() uint64    { return uint64(C.gsdf_hip_evaluations(s.h)) }
func (s *SDF2HIP) Handle() unsafe.Pointer { return unsafe.Pointer(s.h) }
func (s *SDF2HIP) Close() {
	if s.h != nil {
		C.gsdf_hip_program_destroy(s.h)
		s.h = nil
	}
}

func (s *SDF2HIP) Evaluate(pos []ms2.Vec, dist []float32, userData any) error {
	var pp, dp unsafe.Pointer
	if len(pos) > 0 {
		pp = unsafe.Pointer(&pos[0])
	}
	if len(dist) > 0 {
		dp = unsafe.Pointer(&dist[0])
	}
	rc := C.gsdf_hip_eval2(s.h, pp, C.size_t(unsafe.Sizeof(ms2.Vec{})), C.size_t(len(pos)), (*C.float)(dp), C.size_t(len(dist)))
	if rc != 0 {
		return hipErr(rc)
	}
	return nil
}

// NormalsCentralDiffHIP is NormalsCentralDiff (gleval.go:53-108) in one launch: 6 evaluations per point on device.
func (s *SDF3HIP) NormalsCentralDiffHIP(pos []ms3.Vec, normals []ms3.Vec, step float32) error {
	if len(pos) != len(normals) {
		return errMismatchBufferLength
	}
	if len(pos) == 0 {
		return errEmptyBuffers
	}
	rc := C.gsdf_hip_normals3(s.h, (*C.float)(unsafe.Pointer(&pos[0])), (*C.float)(unsafe.Pointer(&normals[0])), C.size_t(len(pos)), C.float(step))
	if rc != 0 {
		return hipErr(rc)
	}
	return nil
}
