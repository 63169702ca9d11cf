"""ctypes mirrors of include/gsdf_program.h (data layout only)."""
import ctypes as C

NPARAM = 8


class GsdfNode(C.Structure):
    _fields_ = [("op", C.c_uint16), ("nchild", C.c_uint16), ("link_off", C.c_uint32),
                ("aux_off", C.c_uint32), ("aux_len", C.c_uint32), ("p", C.c_float * NPARAM)]


class GsdfTree(C.Structure):
    _fields_ = [("nodes", C.POINTER(GsdfNode)), ("n_nodes", C.c_uint32),
                ("links", C.POINTER(C.c_uint32)), ("n_links", C.c_uint32),
                ("aux", C.POINTER(C.c_float)), ("n_aux", C.c_uint32),
                ("root", C.c_uint32), ("bb", C.c_float * 6)]


assert C.sizeof(GsdfNode) == 48

# enum gsdf_op (include/gsdf_program.h) -- order must match the header.
OPS = ["INVALID", "SPHERE", "BOX", "BOXFRAME", "TORUS", "CYLINDER", "HEX", "UNION", "INTERSECT", "DIFF", "XOR",
       "SMOOTH_UNION", "SMOOTH_DIFF", "SMOOTH_INTERSECT", "SCALE", "SYMMETRY", "ARRAY", "ELONGATE", "SHELL",
       "OFFSET", "TRANSLATE", "TRANSFORM", "CIRCARRAY", "TWIST", "EXTRUSION", "REVOLUTION", "SCREW",
       "LINE2D", "ARC2D", "QUADBEZIER2D", "CIRCLE2D", "EQTRI2D", "RECT2D", "DIAMOND2D", "X2D", "HEX2D", "OCT2D",
       "ELLIPSE2D", "POLY2D", "LINES2D", "UNION2D", "INTERSECT2D", "DIFF2D", "XOR2D", "ARRAY2D", "OFFSET2D",
       "TRANSLATE2D", "SYMMETRY2D", "ANNULUS2D", "CIRCARRAY2D", "TRANSLATEMULTI2D", "ROTATION2D", "SCALE2D",
       "ELONGATE2D"]
OP = {n: i for i, n in enumerate(OPS)}
FIRST_2D = OP["LINE2D"]
