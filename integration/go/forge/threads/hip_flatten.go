package threads

// hip_flatten.go -- goes into github.com/soypat/gsdf/forge/threads. The one node type outside package gsdf:
// screw (threads.go:62-69), evaluated at threads.go:141-181. Fields verbatim; lead is already -pitch*starts and
// lengthDiv2 already halved (threads.go:85-95).

import (
	"github.com/soypat/gsdf/glbuild"
	"github.com/soypat/gsdf/gleval"
)

func (s *screw) AppendHIPNodes(f *gleval.HIPFlattener) (uint32, error) {
	return f.Op2(gleval.HIPScrew, []glbuild.Shader2D{s.thread}, nil, s.pitch, s.lead, s.lengthDiv2, s.taper)
}
