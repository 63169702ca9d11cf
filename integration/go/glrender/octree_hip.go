//go:build cgo && hip

package glrender

// octree_hip.go -- goes into github.com/soypat/gsdf/glrender; built with `-tags hip`. Renderers over the device meshers of
// libgsdfhip.so: every octree level, every evaluation and marching cubes (or the flat lattice, or dual contouring) run
// on the GPU inside ONE library call; ReadTriangles only drains the result, so RenderAll (glrender.go:17-36) and
// WriteBinarySTL (stl.go:15-62) work unchanged on top of it.

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/gsdf_amd/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/gsdf_amd/gsdf_amd/csrc -lgsdfhip
#include <stdlib.h>
#include "gsdf_hip.h"
*/
import "C"

import (
	"errors"
	"io"
	"unsafe"

	"github.com/soypat/geometry/ms3"
	"github.com/soypat/gsdf/gleval"
)

func hipErr(rc C.int) error { return errors.New("gsdf_hip: " + C.GoString(C.gsdf_hip_last_error())) }

func hipProgram(s *gleval.SDF3HIP) *C.gsdf_program { return (*C.gsdf_program)(s.Handle()) }

// MeshHIP is a mesh resident on the device, drained through ReadTriangles. It implements Renderer (glrender.go:11-13).
type MeshHIP struct {
	m      *C.gsdf_mesh
	st     C.gsdf_mesh_stats
	n, cur uint64
}

func newMeshHIP(m *C.gsdf_mesh) *MeshHIP {
	o := &MeshHIP{m: m}
	C.gsdf_hip_mesh_stats_get(m, &o.st)
	o.n = uint64(o.st.n_tris)
	return o
}

// NewOctreeRendererHIP stands where NewOctreeRenderer does (octreerenderer.go:43-61; gsdfaux/gsdfaux.go:170). The
// evaluation buffer size of the reference has no meaning on the device. Errors carry the reference's texts
// ("invalid renderer cube resolution", "resolution not fine enough ...").
func NewOctreeRendererHIP(s *gleval.SDF3HIP, cubeResolution float32) (*MeshHIP, error) {
	opts := C.gsdf_mesh_opts{prune: 1, shard_rank: 0, shard_count: 1} // payload 0 = triangles, share_corners 0 = every corner of every leaf, as marchCubes (marchcubes.go:24-31)
	var m *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_octree(hipProgram(s), C.float(cubeResolution), &opts, &m); rc != 0 {
		return nil, hipErr(rc)
	}
	return newMeshHIP(m), nil
}

// NewFlatRendererHIP mirrors FlatRenderer.Reset(s, cubeResolution, evalBufferSize, numParallel) (flatrenderer.go:36-70):
// the two buffer arguments keep their validation (flatrenderer.go:37-45) and have no other meaning on the device.
func NewFlatRendererHIP(s *gleval.SDF3HIP, cubeResolution float32, evalBufferSize, numParallel int) (*MeshHIP, error) {
	if evalBufferSize < 8 {
		return nil, errors.New("flat renderer eval buffer size must be at least 8")
	}
	if numParallel < 1 {
		return nil, errors.New("flat renderer numParallel must be at least 1")
	}
	var m *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_flat(hipProgram(s), C.float(cubeResolution), 0, 1, nil, &m); rc != 0 {
		return nil, hipErr(rc)
	}
	return newMeshHIP(m), nil
}

// NewDualContourRendererHIP: DualContourRenderer.Reset + DualContourLeastSquares (dual_contour.go:26-293,
// dual_contour_vertexplacement.go:26-223) on device.
func NewDualContourRendererHIP(s *gleval.SDF3HIP, cubeResolution float32, chiseled bool) (*MeshHIP, error) {
	ch := C.int(0)
	if chiseled {
		ch = 1
	}
	var m *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_dualcontour(hipProgram(s), C.float(cubeResolution), ch, 0, 1, nil, &m); rc != 0 {
		return nil, hipErr(rc)
	}
	return newMeshHIP(m), nil
}

// NewMinecraftRendererHIP: minecraftRender (dual_contour.go:297-403; unexported there) -- the axis-aligned faces between level-1
// cubes whose origins lie on different sides of the surface.
func NewMinecraftRendererHIP(s *gleval.SDF3HIP, cubeResolution float32) (*MeshHIP, error) {
	var m *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_minecraft(hipProgram(s), C.float(cubeResolution), nil, &m); rc != 0 {
		return nil, hipErr(rc)
	}
	return newMeshHIP(m), nil
}

// ReadTriangles: the iterator contract of octreerenderer.go:131-134,154-157 -- len(dst) >= 5 or io.ErrShortBuffer,
// (n, nil) while more remain, (n, io.EOF) at the end.
func (o *MeshHIP) ReadTriangles(dst []ms3.Triangle, userData any) (int, error) {
	if len(dst) < 5 {
		return 0, io.ErrShortBuffer
	}
	n := min(uint64(len(dst)), o.n-o.cur)
	if n > 0 {
		if rc := C.gsdf_hip_mesh_read(o.m, C.uint64_t(o.cur), C.uint64_t(n), (*C.float)(unsafe.Pointer(&dst[0]))); rc != 0 {
			return 0, hipErr(rc)
		}
		o.cur += n
	}
	if o.cur == o.n {
		return int(n), io.EOF
	}
	return int(n), nil
}

// TotalPruned is Octree.TotalPruned (octreerenderer.go:279-284): leaves discarded by the centre tests.
func (o *MeshHIP) TotalPruned() uint64 { return uint64(o.st.pruned_leaves) }

// Evaluations performed on device for this mesh (FlatRenderer: (nx+1)(ny+1)(nz+1)).
func (o *MeshHIP) Evaluations() uint64 { return uint64(o.st.evals) }

// NumTriangles in the mesh.
func (o *MeshHIP) NumTriangles() uint64 { return o.n }

// Close frees the device (and pinned host) buffers of the mesh; views returned by Triangles / STL die with it.
func (o *MeshHIP) Close() {
	if o.m != nil {
		C.gsdf_hip_mesh_destroy(o.m)
		o.m = nil
	}
}

// WriteBinarySTL of a device mesh (stl.go:15-62 byte for byte: 80-byte zero header, count, 50-byte records with
// Unit(Normal())): the records are built on device and the whole file arrives by one DMA in pinned host memory the
// mesh owns, so the Go side is ONE Write (the reference issues one per triangle, stl.go:53).
func (o *MeshHIP) WriteBinarySTL(w io.Writer) (int, error) {
	var p *C.uint8_t
	var n C.size_t
	if rc := C.gsdf_hip_mesh_host_stl(o.m, &p, &n); rc != 0 {
		return 0, hipErr(rc) // "empty triangle slice", "amount of triangles in model exceeds STL design limits"
	}
	return w.Write(unsafe.Slice((*byte)(unsafe.Pointer(p)), int(n)))
}

// Triangles without the ReadTriangles loop (RenderAll appends 4096 at a time into pageable memory): a read-only
// view of pinned host memory filled by one DMA, valid until Close.
func (o *MeshHIP) Triangles() ([]ms3.Triangle, error) {
	if o.n == 0 {
		return nil, nil
	}
	var p *C.float
	if rc := C.gsdf_hip_mesh_host_tris(o.m, &p); rc != 0 {
		return nil, hipErr(rc)
	}
	return unsafe.Slice((*ms3.Triangle)(unsafe.Pointer(p)), int(o.n)), nil
}

// ---- several GPUs of one node: one process per GPU, bricks dealt by a coordinate hash, one gather at the end ----

// HIPUniqueID: rank 0 draws the 128-byte communicator id and ships it to the other ranks (file, socket, environment).
func HIPUniqueID() (id [C.GSDF_COMM_ID_BYTES]byte, err error) {
	if rc := C.gsdf_hip_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		return id, hipErr(rc)
	}
	return id, nil
}

// HIPComm is the library's RCCL communicator (one rank per GPU over xGMI).
type HIPComm struct{ h *C.gsdf_comm }

// NewHIPComm is collective: every rank calls it after gleval.InitHIP(localDevice).
func NewHIPComm(id [C.GSDF_COMM_ID_BYTES]byte, rank, world int) (*HIPComm, error) {
	var h *C.gsdf_comm
	if rc := C.gsdf_hip_comm_create((*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(rank), C.int(world), &h); rc != 0 {
		return nil, hipErr(rc)
	}
	return &HIPComm{h}, nil
}

func (c *HIPComm) Close() { C.gsdf_hip_comm_destroy(c.h) }

// SumUint64 adds vals over the ranks in place (Evaluations(), TotalPruned() of the whole job).
func (c *HIPComm) SumUint64(vals []uint64) error {
	if len(vals) == 0 {
		return nil
	}
	if rc := C.gsdf_hip_comm_allreduce_sum_u64(c.h, (*C.uint64_t)(unsafe.Pointer(&vals[0])), C.size_t(len(vals))); rc != 0 {
		return hipErr(rc)
	}
	return nil
}

// NewOctreeShardHIP meshes this rank's share of the bricks and keeps it as packed cut-leaf records (40 B per cut
// leaf = 20 B per triangle): what to hand to Gather -- the receiving ranks run marching cubes behind the transfer.
func NewOctreeShardHIP(s *gleval.SDF3HIP, cubeResolution float32, rank, world int) (*MeshHIP, error) {
	opts := C.gsdf_mesh_opts{prune: 1, shard_rank: C.int(rank), shard_count: C.int(world), payload: C.GSDF_PAYLOAD_RECORDS}
	var m *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_octree(hipProgram(s), C.float(cubeResolution), &opts, &m); rc != 0 {
		return nil, hipErr(rc)
	}
	return newMeshHIP(m), nil // no triangles yet: Gather it, or March to keep the shard here
}

// March turns a records payload into triangles in place (a shard that stays on its rank).
func (o *MeshHIP) March() error {
	if rc := C.gsdf_hip_mesh_march(o.m); rc != 0 {
		return hipErr(rc)
	}
	C.gsdf_hip_mesh_stats_get(o.m, &o.st)
	o.n = uint64(o.st.n_tris)
	return nil
}

// Gather is collective: the result holds the triangles of ALL ranks and reads like any MeshHIP; counts[r] = rank r's.
func (o *MeshHIP) Gather(c *HIPComm) (*MeshHIP, []uint64, error) {
	counts := make([]uint64, int(C.gsdf_hip_comm_world(c.h)))
	var all *C.gsdf_mesh
	if rc := C.gsdf_hip_mesh_gatherv(o.m, c.h, &all, (*C.uint64_t)(unsafe.Pointer(&counts[0]))); rc != 0 {
		return nil, nil, hipErr(rc)
	}
	return newMeshHIP(all), counts, nil
}

// PendingGatherHIP is a gather whose counts are exchanged and whose payload is enqueued on the communicator's stream.
type PendingGatherHIP struct {
	g     *C.gsdf_gather
	world int
}

// GatherStart returns as soon as everything is enqueued: mesh the next part before Wait. mode: C.GSDF_GATHER_ALL (0) every
// rank receives, GSDF_GATHER_ROOT (1) only root, GSDF_GATHER_NONE (2) counts only. The source mesh may be closed at once.
func (o *MeshHIP) GatherStart(c *HIPComm, mode, root int) (*PendingGatherHIP, error) {
	var g *C.gsdf_gather
	if rc := C.gsdf_hip_mesh_gatherv_start(o.m, c.h, C.int(mode), C.int(root), &g); rc != 0 {
		return nil, hipErr(rc)
	}
	return &PendingGatherHIP{g, int(C.gsdf_hip_comm_world(c.h))}, nil
}

// Wait blocks until the gathered mesh is complete; nil mesh on a rank that receives nothing (mode root / none).
func (p *PendingGatherHIP) Wait() (*MeshHIP, []uint64, error) {
	counts := make([]uint64, p.world)
	var all *C.gsdf_mesh
	var gst C.gsdf_gather_stats
	if rc := C.gsdf_hip_mesh_gatherv_wait(p.g, &all, (*C.uint64_t)(unsafe.Pointer(&counts[0])), &gst); rc != 0 {
		return nil, nil, hipErr(rc)
	}
	if all == nil {
		return nil, counts, nil
	}
	return newMeshHIP(all), counts, nil
}
