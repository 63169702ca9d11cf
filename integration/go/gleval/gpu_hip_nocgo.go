//go:build !cgo || !hip

package gleval

// gpu_hip_nocgo.go -- goes into github.com/soypat/gsdf/gleval: what a build without cgo or without `-tags hip` sees
// (the pattern of gpu_nocgo.go:1, `//go:build tinygo || !cgo`): the names exist, the constructors fail.

import (
	"errors"
	"unsafe"

	"github.com/soypat/geometry/ms2"
	"github.com/soypat/geometry/ms3"
)

var errNoHIP = errors.New("gsdf_hip: built without cgo or without the hip build tag")

type HIPConfig struct{ Specialize, Wait bool }

func InitHIP(device int) (terminate func(), err error) { return nil, errNoHIP }

type SDF3HIP struct{ bb ms3.Box }

func NewHIPSDF3(t HIPTree, bb ms3.Box, cfg HIPConfig) (*SDF3HIP, error) { return nil, errNoHIP }
func (s *SDF3HIP) Bounds() ms3.Box                                       { return s.bb }
func (s *SDF3HIP) Evaluations() uint64                                   { return 0 }
func (s *SDF3HIP) Handle() unsafe.Pointer                                { return nil }
func (s *SDF3HIP) Close()                                                {}
func (s *SDF3HIP) Evaluate(pos []ms3.Vec, dist []float32, userData any) error {
	return errNoHIP
}

type SDF2HIP struct{ bb ms2.Box }

func NewHIPSDF2(t HIPTree, bb ms2.Box, cfg HIPConfig) (*SDF2HIP, error) { return nil, errNoHIP }
func (s *SDF2HIP) Bounds() ms2.Box                                       { return s.bb }
func (s *SDF2HIP) Evaluations() uint64                                   { return 0 }
func (s *SDF2HIP) Handle() unsafe.Pointer                                { return nil }
func (s *SDF2HIP) Close()                                                {}
func (s *SDF2HIP) Evaluate(pos []ms2.Vec, dist []float32, userData any) error {
	return errNoHIP
}
