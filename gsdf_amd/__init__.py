"""gsdf_amd -- MI355X-native SDF evaluation + meshing backend behind soypat/gsdf's gleval.SDF3 /
glrender.Renderer interfaces. See DESIGN.md. The HIP library is loaded lazily by gsdf_amd.hip and
fails loudly if missing; nothing here falls back to a CPU path."""
