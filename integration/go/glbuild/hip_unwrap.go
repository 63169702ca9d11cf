package glbuild

// hip_unwrap.go -- goes into github.com/soypat/gsdf/glbuild. The one line the HIP flattener needs from this package:
// unwraproot (glbuild.go:1366-1375) strips the decorators that forward evaluation to the node they wrap
// (nameOverloadShader3D/2D, CachedShader3D/2D, overloadBounds3/2: glbuild.go:1087-1333); their unwrap() methods are
// unexported, so a walker outside this package cannot see through them.

// UnwrapHIP returns the innermost shader of a chain of glbuild decorators (s itself if it is none).
func UnwrapHIP(s Shader) Shader {
	if s == nil {
		return nil
	}
	return unwraproot(s)
}
