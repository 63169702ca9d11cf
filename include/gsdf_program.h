/*
 * gsdf_program.h -- flattened gsdf CSG tree ("tree blob"): the DATA format that crosses the C ABI.
 *
 * One gsdf_node per reference node; the fields are the reference node structs' own (unexported)
 * fields, verbatim, so that the Go-side flattener is a field copy. Anything
 * derived (cylinder args(), 1/scale, polygon edge constants ...) is computed by the consumer.
 *
 * Reference node structs (file:line are relative to /root/reference):
 *   3D prims  primitives.go:23,75,119,147,164,207,266     3D ops   operations.go:27-848
 *   2D prims  primitives2d.go:33-640                       2D ops   operations2d.go:15-819
 *   screw     forge/threads/threads.go:62-69
 *
 * Children are indices into the same node array, stored in a separate `links` array
 * (node.link_off .. node.link_off+node.nchild). Variable-length float payloads (polygon vertices,
 * line segments, displacement lists, 4x4 / 2x2 matrices) live in the `aux` float pool.
 */
#ifndef GSDF_PROGRAM_H
#define GSDF_PROGRAM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum gsdf_op {
  GSDF_OP_INVALID = 0,
  /* ---- 3D primitives (cpu_evaluators.go:20-105) ---- */
  GSDF_SPHERE,        /* p0=r */
  GSDF_BOX,           /* p0..2=dims p3=round */
  GSDF_BOXFRAME,      /* p0..2=dims p3=e (already halved by the builder, primitives.go:255) */
  GSDF_TORUS,         /* p0=rGreater p1=rLesser */
  GSDF_CYLINDER,      /* p0=r p1=h p2=round */
  GSDF_HEX,           /* p0=side p1=h */
  /* ---- 3D booleans (cpu_evaluators.go:124-286) ---- */
  GSDF_UNION,         /* n-ary, nchild>=2 */
  GSDF_INTERSECT,
  GSDF_DIFF,
  GSDF_XOR,
  GSDF_SMOOTH_UNION,     /* p0=k */
  GSDF_SMOOTH_DIFF,      /* p0=k */
  GSDF_SMOOTH_INTERSECT, /* p0=k */
  /* ---- 3D unary ops (cpu_evaluators.go:288-504,1042,1257) ---- */
  GSDF_SCALE,         /* p0=scale */
  GSDF_SYMMETRY,      /* p0 = bit mask as float (1=x 2=y 4=z) */
  GSDF_ARRAY,         /* p0..2=spacing p3..5=nx,ny,nz (as float) */
  GSDF_ELONGATE,      /* p0..2=h */
  GSDF_SHELL,         /* p0=thick */
  GSDF_OFFSET,        /* p0=off */
  GSDF_TRANSLATE,     /* p0..2=t */
  GSDF_TRANSFORM,     /* aux[0..15]=tInv row-major x00,x01,..x33 (inverse computed by the builder) */
  GSDF_CIRCARRAY,     /* p0=nInst p1=circleDiv (as float) */
  GSDF_TWIST,         /* p0=k */
  /* ---- 2D -> 3D (cpu_evaluators.go:506-549, threads.go:141-181) ---- */
  GSDF_EXTRUSION,     /* p0=h ; child is 2D */
  GSDF_REVOLUTION,    /* p0=off ; child is 2D */
  GSDF_SCREW,         /* p0=pitch p1=lead p2=lengthDiv2 p3=taper ; child is 2D */
  /* ---- 2D primitives (cpu_evaluators.go:551-818,1145) ---- */
  GSDF_LINE2D,        /* p0,1=a p2,3=b p4=width */
  GSDF_ARC2D,         /* p0=radius p1=angle p2=thick */
  GSDF_QUADBEZIER2D,  /* p0,1=a p2,3=b p4,5=c p6=thick */
  GSDF_CIRCLE2D,      /* p0=r */
  GSDF_EQTRI2D,       /* p0=hTri */
  GSDF_RECT2D,        /* p0,1=d */
  GSDF_DIAMOND2D,     /* p0,1=d */
  GSDF_X2D,           /* p0=dim p1=thick */
  GSDF_HEX2D,         /* p0=side */
  GSDF_OCT2D,         /* p0=c */
  GSDF_ELLIPSE2D,     /* p0=a p1=b */
  GSDF_POLY2D,        /* aux = x0,y0,x1,y1,... (aux_len = 2*nverts) */
  GSDF_LINES2D,       /* p0=width ; aux = ax,ay,bx,by per segment (aux_len = 4*nseg) */
  /* ---- 2D ops (cpu_evaluators.go:821-1255) ---- */
  GSDF_UNION2D,
  GSDF_INTERSECT2D,
  GSDF_DIFF2D,
  GSDF_XOR2D,
  GSDF_ARRAY2D,       /* p0,1=spacing p2,3=nx,ny */
  GSDF_OFFSET2D,      /* p0=f */
  GSDF_TRANSLATE2D,   /* p0,1=t */
  GSDF_SYMMETRY2D,    /* p0=bit mask (1=x 2=y) */
  GSDF_ANNULUS2D,     /* p0=r */
  GSDF_CIRCARRAY2D,   /* p0=nInst p1=circleDiv */
  GSDF_TRANSLATEMULTI2D, /* aux = dx,dy per displacement */
  GSDF_ROTATION2D,    /* p0..3 = tInv x00,x01,x10,x11 */
  GSDF_SCALE2D,       /* p0=scale */
  GSDF_ELONGATE2D,    /* p0,1=h */
  GSDF_OP_COUNT
};

#define GSDF_NODE_NPARAM 8

typedef struct gsdf_node {
  uint16_t op;        /* enum gsdf_op */
  uint16_t nchild;    /* number of children */
  uint32_t link_off;  /* first child slot in links[] */
  uint32_t aux_off;   /* first float in aux[] */
  uint32_t aux_len;   /* number of floats in aux[] */
  float    p[GSDF_NODE_NPARAM];
} gsdf_node;          /* 48 bytes */

/* A whole tree. All pointers are borrowed for the duration of the call that receives them. */
typedef struct gsdf_tree {
  const gsdf_node* nodes;
  uint32_t         n_nodes;
  const uint32_t*  links;
  uint32_t         n_links;
  const float*     aux;
  uint32_t         n_aux;
  uint32_t         root;      /* index of the root node (must be a 3D node for eval3, 2D for eval2) */
  float            bb[6];     /* Bounds(): min xyz, max xyz (2D: min xy 0, max xy 0) */
} gsdf_tree;

/* The layout cgo sees (Go mirrors C struct layout field by field; INTEGRATION.md section 1 fills these structs from Go):
 * pinned here so that a change of this header that would silently break a Go caller breaks the build instead. */
#ifdef __cplusplus
#define GSDF_ABI_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define GSDF_ABI_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
#include <stddef.h>
GSDF_ABI_ASSERT(sizeof(float) == 4 && sizeof(void*) == 8, "LP64, IEEE float32");
GSDF_ABI_ASSERT(sizeof(gsdf_node) == 48, "gsdf_node is 48 bytes");
GSDF_ABI_ASSERT(offsetof(gsdf_node, op) == 0 && offsetof(gsdf_node, nchild) == 2 && offsetof(gsdf_node, link_off) == 4, "gsdf_node head");
GSDF_ABI_ASSERT(offsetof(gsdf_node, aux_off) == 8 && offsetof(gsdf_node, aux_len) == 12 && offsetof(gsdf_node, p) == 16, "gsdf_node tail");
GSDF_ABI_ASSERT(sizeof(gsdf_tree) == 72, "gsdf_tree is 72 bytes");
GSDF_ABI_ASSERT(offsetof(gsdf_tree, nodes) == 0 && offsetof(gsdf_tree, n_nodes) == 8 && offsetof(gsdf_tree, links) == 16, "gsdf_tree head");
GSDF_ABI_ASSERT(offsetof(gsdf_tree, n_links) == 24 && offsetof(gsdf_tree, aux) == 32 && offsetof(gsdf_tree, n_aux) == 40, "gsdf_tree middle");
GSDF_ABI_ASSERT(offsetof(gsdf_tree, root) == 44 && offsetof(gsdf_tree, bb) == 48, "gsdf_tree tail");
GSDF_ABI_ASSERT(GSDF_SPHERE == 1 && GSDF_UNION == 7 && GSDF_SCREW == 26 && GSDF_POLY2D == 38 && GSDF_OP_COUNT == 54, "gsdf_op numbering is part of the ABI");

/* 1 if op takes 2D positions. */
static inline int gsdf_op_is2d(int op) { return op >= GSDF_LINE2D && op < GSDF_OP_COUNT; }

#ifdef __cplusplus
}
#endif
#endif
