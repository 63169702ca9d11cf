// dev_ops.h -- device instruction set of the gsdf_hip SDF interpreter (product-internal).
//
// A gsdf tree blob (include/gsdf_program.h) is lowered on the host (compile.cpp) to a straight-line
// u32 instruction stream that every lane executes in lock step (wave-uniform control flow: the
// program counter and every parameter live in SGPRs). Per-lane state is the current position
// P (3 VGPRs), the current distance R (1 VGPR) and a per-lane scratch column in LDS
// ("LDS-staged node stack", slot s of lane t at lds[s*blockDim + t], conflict-free).
// Slots are allocated statically by the host compiler, so there is no stack pointer at run time.
//
// Encoding: word0 = opcode (bits 0-11) | flags (bits 12-15) | slot<<16 ; then `nparam` 32-bit words.
// The stream is straight-line except for D_GATE*, a forward wave-uniform skip. Read-only tables referenced by
// instructions (D_CIRC_PRE) follow D_END.
//   D_FLAG_HXY  (bit 14): hypot(P.x,P.y) of every point is already in the per-point `hxy` register (the host
//                compiler proved P.xy unchanged since it was last computed): reuse instead of recomputing.
//   D_FLAG_SHXY (bit 13): P.x and P.y are functions of the ENTRY x,y only (no z has been mixed in), and
//   D_FLAG_SHZ  (bit 12): P.z is a function of the entry z only. The mesher's leaf kernels feed the interpreter
//                the corners of one leaf cube in the order {0,4,1,5 | 3,7,2,6}: points 2j and 2j+1 enter with the
//                same x,y and (4 points per lane) points j and j+2 with the same z, so the flagged instruction's
//                f(P.x,P.y) (hypot, atan2) resp. g(P.z) (twist sin/cos) is computed once per pair and copied --
//                same inputs, same operations, same bits. Ignored by every other kernel (arbitrary positions).
//   D_FLAG_SWAP (bit 15): combine with operand roles exchanged (the second child was evaluated first so that
//                the position did not have to be saved and restored).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define D_OP_MASK 0x0fffu
#define D_FLAG_SHZ 0x1000u
#define D_FLAG_SHXY 0x2000u
#define D_FLAG_HXY 0x4000u
#define D_FLAG_SWAP 0x8000u

enum DevOp : uint32_t {
  D_END = 0,
  // ---- 3D primitives: R = f(P)
  D_SPHERE,    // r
  D_BOX,       // hx hy hz round            (h = 0.5*dims)
  D_BOXFRAME,  // e bx by bz
  D_TORUS,     // R r
  D_CYL0,      // r h'                       (round == 0 path)
  D_CYLR,      // r h' round
  D_HEX,       // h1 h2 clm
  // ---- 2D primitives: R = f(P.xy)
  D_LINE2D,    // ax ay bax bay dotba w
  D_ARC2D,     // r t s c scrx scry
  D_QUADBEZIER2D,  // Ax Ay ax ay a2 bx by cx cy kk kx kx2 thick
  D_CIRCLE2D,  // r
  D_EQTRI2D,   // r r/k
  D_RECT2D,    // bx by
  D_DIAMOND2D, // bx by dotbb hbx hby bxby
  D_X2D,       // w r
  D_HEX2D,     // r kzr
  D_OCT2D,     // r kzr
  D_ELLIPSE2D, // a b
  D_POLY2D,    // nv v0x v0y, pad to an 8-dword boundary, then nv x {v1x v1y ex ey n2e v2y RN(1/n2e)|0 0}
  D_LINES2D,   // ns w then ns x {ax ay bax bay dotba}
  // ---- position pre-ops: P = T(P)
  D_TRANSLATE,     // tx ty tz
  D_SCALE_PRE,     // inv
  D_SYMMETRY,      // bits
  D_TRANSFORM,     // m00 m01 m02 m03 m10 .. m23 (12)
  D_TWIST,         // k
  D_ROT2D,         // x00 x01 x10 x11
  D_EXTRUDE_PRE,   // h/2            slot <- |z|-h/2
  D_REVOLVE_PRE,   // off
  D_SCREW_PRE,     // pitch lead L tanTaper halfpitch RN(1/pitch)|0   slot <- |z|-L
  D_ELONGATE_PRE,  // hx hy hz       slot <- min(max3(q),0)
  D_ELONGATE2D_PRE,// hx hy          slot <- min(max2(q),0)
  D_ARRAY_PRE,     // i j k sx sy sz nx ny nz   P = f(saved P at slot..slot+2)
  D_ARRAY2D_PRE,   // i j sx sy nx ny           P = f(saved P at slot..slot+1)
  D_CIRC_PRE,      // angle ncirc ninsm1 tab : P = p1 ; slot..slot+1 <- p0.xy (3D keeps z in place). tab: word offset of
                   // the host-computed {sin, cos}(angle * i), i < ncirc, stored after D_END (0: compute on device)
  D_LOADP2_SUB,    // dx dy          P.xy = saved(slot..slot+1) - d
  // ---- distance post-ops: R = g(R)
  D_MULR,          // f
  D_SHELL_POST,    // th
  D_ADDR,          // off
  D_ANNULUS,       // r
  D_EXTRUDE_POST,  // (slot)
  D_MAXR_SLOT,     // (slot)  R = max(R, lds[slot])
  D_ADDR_SLOT,     // (slot)  R += lds[slot]
  // ---- scratch
  D_SAVEP3, D_LOADP3, D_SAVEP2, D_LOADP2, D_SAVER,  // (slot)
  D_SETSLOT,       // value (slot)
  D_SETR,          // value : R = value
  // ---- combine: a = lds[slot] (first operand), b = R  ->  R
  D_COMBINE_MIN, D_COMBINE_MAX, D_COMBINE_DIFF, D_COMBINE_XOR,
  D_COMBINE_SUNION, D_COMBINE_SDIFF, D_COMBINE_SINTER,  // k RN(1/k)|0
  // ---- gates: skip a child that provably cannot influence its parent's combine for ANY point of the wave.
  //      a = lds[slot] is the value of the children evaluated so far; the child's field is bounded from below, outside
  //      the child's region, by L(p) (compile.cpp: lower_region; box: Chebyshev distance, z-cylinder: max(z excess,
  //      rs * radial excess, rs * (rin - rho))). If for every point of the wave L > 0 and L > sg * a + kk + margin (sg = +1: union and
  //      smooth union, -1: difference and smooth difference; kk = 1.002 k for the smooth combines, else 0; margin =
  //      1e-3 (L + |a|) + 2e-6 (|x| + |y| + |z|), a thousand times the rounding of either side), the child's value b >= L
  //      leaves the combine's result unchanged bit for bit (see gen_combine): R = L and the program counter advances
  //      by `skip` words to the child's combine instruction, which runs on (a, L). Wave-uniform forward branch -- the one
  //      non-straight-line instruction of the stream. The skip word's top byte holds the child's brick-mask number + 1
  //      (0: none; see D_SKIP below).
  //      Context of ONE enclosing combine (oslot != 0xffff; differences only): the gated child b is the subtrahend of
  //      X = (smooth) max(a, -b), and X is itself the minuend of an enclosing (smooth) difference max(X, -c) whose other
  //      operand c was evaluated first and sits in lds[oslot]. With U = max(a, -L) + k4 >= X (k4 = a quarter of the inner
  //      blend width, 0 for a plain difference): if U < 0 and c + U <= -ok - margin (ok = 1.002 x the outer blend width, 0
  //      for a plain difference), the enclosing combine yields -c whatever X is (its weight clamps; X only enters times 0,
  //      and substitute and true value are both negative) -- b is skipped although it does change X. A lane passes if
  //      either test holds. Example: a body with cutters, then a through hole (knurled-cylinder): at the hole's wall the
  //      cutters change the body's field but the hole discards it.
  D_GATE2D,  // minx miny maxx maxy sg kk oslot ok k4 skip
  D_GATE3D,  // minx miny minz maxx maxy maxz sg kk oslot ok k4 skip
  D_GATEZC,  // cx cy r z0 z1 rs rin sg kk oslot ok k4 skip (D_FLAG_HXY: cx = cy = 0 and hypot(P.x,P.y) is in the register)
  //      D_UBOUND* opens a wide union: lds[slot] <- (1 + 1e-3) * min over the listed boxes of the distance to the box's
  //      FARTHEST corner -- an upper bound of that child's field (the shape lies inside its box), hence of the union.
  //      Starting the running minimum there lets the gates drop far children from the first one on; the bound never
  //      reaches the result (it is >= the nearest child's value, which is always evaluated: its L <= bound).
  D_UBOUND2D,   // nb then nb x {minx miny maxx maxy}
  D_UBOUND3D,   // nb then nb x {minx miny minz maxx maxy maxz}
  //      D_GATEOB: the same gate for a region that is a box turned about z (compile.cpp: Region::OBOX -- what a rotation about
  //      z makes of a box; an axis-aligned hull of it would be up to sqrt 2 wider): (c, s) = the box's own x axis in the
  //      current frame, L = max(|c dx + s dy| - hx, |c dy - s dx| - hy, z0 - z, z - z1) with (dx, dy) = P.xy - (cx, cy),
  //      the Chebyshev distance in the box's frame, a lower bound of the Euclidean one.
  //      D_CIRC_ORDER (after D_CIRC_PRE, when the array's child has such a region): a circular array evaluates its child in
  //      the point's own sector and in the next one and keeps the minimum (cpu_evaluators.go:1082-1090) -- min is
  //      commutative bit for bit, so the wave may start with whichever copy is nearer: if for most of its lanes the copy at
  //      lds[slot..slot+1] (p0) lies nearer the child's region than the one in P (p1), the two change places. The farther
  //      copy then comes second, behind a D_GATEOB against the value of the first (the union rule: L > a).
  D_GATEOB,     // cx cy c s hx hy z0 z1 sg kk oslot ok k4 skip
  D_CIRC_ORDER, // cx cy c s hx hy   (slot = the D_CIRC_PRE's)
  // ---- interval mode (sdf_eval<2, 0, LIP = true>; prune_kernel only -- every other kernel steps over these four).
  //      The octree drops a cube when the field cannot vanish inside it. The reference decides that from the centre value
  //      alone, |d| >= size * sqrt3/2 (octreerenderer.go:270-273): right for true distance fields, wrong for fields that grow
  //      faster than distance (twist, screw, non-rigid transform) or jump (a screw with an asymmetric thread form, across
  //      the seams of its sawtooth) -- examples/fibonacci-showerhead loses 23 triangles to it. In interval mode the lane's
  //      two "points" are the SAME cube centre; point 0 carries a lower bound and point 1 an upper bound of the field over
  //      the cube's bounding ball. lipR = radius of a ball holding that ball's image in the current frame (h at the root,
  //      times the scale factors, times the stretch of the non-1-Lipschitz maps, plus the seam term): primitives, being exact
  //      distances, yield value -+ lipR; monotone instructions act on the two points unchanged; the rest cross them
  //      (difference, xor, |d|). For a tree of primitives under rigid motions and min / max the bounds are d -+ h exactly,
  //      i.e. the reference's predicate. The oracle does the same with doubled batches (oracle/orc_eval.c: lipctx).
  //      Gates work in interval mode too, on the bound over the whole ball: a region's lower bound L is 1-Lipschitz, so the
  //      child is >= L(centre) - lipR everywhere in the ball; the test runs on that against both ends of `a` (the two points),
  //      and the substitute (both ends = L - lipR) leaves the combine's interval what the ungated evaluation gives, bit for
  //      bit (min / max select a's ends; a clamped blend weight multiplies the substitute by 0) -- the oracle has no gates.
  D_LIP_PUSH,   // (slot = nesting depth)  interval stack[depth] <- lipR, |P.x| + |P.y| + |P.z|  (before a map that stretches; at the entry of a combine frame with numbered operands)
  D_LIP_POP,    // (slot = nesting depth)  lipR <- interval stack[depth]           (after its subtree)
  D_LIP_MUL,    // f : lipR *= f  (non-rigid D_TRANSFORM / D_ROT2D: largest singular value, rounded up)
  D_LIP_WRAP,   // halfpitch seam : after D_SCREW_PRE; if |P.x| + lipR >= halfpitch (the image may reach a seam of the
                // sawtooth) lipR += seam (what the profile's field can jump by there; compile.cpp: lip_screw_seam)
  // ---- brick masks (round 5). The last level of centre tests evaluates the field in interval mode over the bounding ball of
  //      every level-3 cube -- exactly the brick (4 x 4 x 4 leaves, 512 corner evaluations) one wave of the leaf kernels then
  //      evaluates point by point. What the intervals prove for the whole ball holds for every corner of the brick, so the
  //      centre test hands the leaf kernel a 16-bit mask with the cube (Cube.w): bit i = "operand subtree i cannot influence
  //      its combine anywhere in this brick". Up to 16 operand subtrees of combine frames (compile.cpp: the most expensive
  //      ones that sit under continuous position maps only -- array cells, circular sectors and screw sawteeth jump, and a
  //      copy's decision at the centre says nothing about the same instruction for a point across the seam) carry a number.
  //      D_SKIP (first instruction of such a subtree): if the wave's brick mask has the bit, R <- subst and the program counter
  //      advances to the end of the subtree (the D_SAVER or combine that follows), a wave-uniform forward branch decided by a
  //      scalar bit test -- no per-point gate test at all. subst = +-1e30, the side of the combine on which the operand
  //      provably lies: min / max then select the other operand, a smooth combine's weight clamps to exactly 0 or 1 and subst
  //      enters times 0 with the sign the true value would have had (a zero result keeps its sign): same bits as evaluating the
  //      subtree. D_FLAG_HXY on D_SKIP: the subtree would have left hypot(P.x, P.y) of the entry position in the register for
  //      code behind it -- the skip path computes it instead. Elsewhere (no brick: Evaluate, lattices, dual contouring; interval
  //      mode itself) the mask is 0 and D_SKIP falls through.
  //      D_LIP_DOM (in front of a combine, interval mode only): records, per lane = per cube, which operand of the combine
  //      dominates over the whole ball, with a = the formula's first operand, b its second (D_FLAG_SWAP: b was evaluated first
  //      and sits in lds[slot], a is in R), kk = 1.002 k for the smooth combines, m = kk + 1e-3 (|u| + |v|) + pad:
  //        kind 0 min(a, b)        : b.lo > a.hi + m -> skip b (+)      a.lo > b.hi + m -> skip a (+)
  //        kind 1 max(a, b)        : b.hi < a.lo - m -> skip b (-)      a.hi < b.lo - m -> skip a (-)
  //        kind 2 max(a, -b)       : b.lo > -a.lo + m -> skip b (+)     a.hi < -b.hi - m -> skip a (-)
  //        kind 3 smooth union     : b.lo - a.hi > m -> skip b (+)      a.lo - b.hi > m -> skip a (+)
  //        kind 4 smooth difference: b.lo + a.lo > m -> skip b (+)      b.hi + a.hi < -m -> skip a (-)
  //        kind 5 smooth intersect.: a.lo - b.hi > m -> skip b (-)      b.lo - a.hi > m -> skip a (-)
  //      ((+) / (-): the sign of the D_SKIP's subst.) pad = 2e-6 (|x| + |y| + |z| + 4 lipR) of the frame's entry position
  //      (pinfo: the depth of the interval stack where the frame's D_LIP_PUSH left it). A D_GATE* in interval mode records its child's bit the same way
  //      (the child's bound outside its region in place of b.lo) for every lane that passes, whether or not the wave as a
  //      whole skips; a gate whose child carries a number is not tested per point by a wave that holds a valid brick mask
  //      (bit 31 of the mask): the brick-level decision replaces it.
  D_SKIP,       // id subst skip|flags<<24 : words from this instruction to the end of the subtree
  D_LIP_DOM,    // kind ida idb kk pinfo   (slot = the combine's; ida / idb = 0xff: none; pinfo = interval-stack depth | is2d << 16)
  D_OP_COUNT
};

// Fixed parameter-word counts (variable: D_POLY2D 3 + pad + 8*nv, D_LINES2D 2 + 5*ns, D_UBOUND2D/3D 1 + 4*nb / 1 + 6*nb).
static const uint8_t kDevOpParams[D_OP_COUNT] = {
    /*END*/ 0,
    /*SPHERE*/ 1, /*BOX*/ 4, /*BOXFRAME*/ 4, /*TORUS*/ 2, /*CYL0*/ 2, /*CYLR*/ 3, /*HEX*/ 3,
    /*LINE2D*/ 6, /*ARC2D*/ 6, /*QUADBEZIER*/ 13, /*CIRCLE*/ 1, /*EQTRI*/ 2, /*RECT*/ 2, /*DIAMOND*/ 6, /*X2D*/ 2,
    /*HEX2D*/ 2, /*OCT2D*/ 2, /*ELLIPSE*/ 2, /*POLY*/ 3, /*LINES*/ 2,
    /*TRANSLATE*/ 3, /*SCALE_PRE*/ 1, /*SYMMETRY*/ 1, /*TRANSFORM*/ 12, /*TWIST*/ 1, /*ROT2D*/ 4,
    /*EXTRUDE_PRE*/ 1, /*REVOLVE_PRE*/ 1, /*SCREW_PRE*/ 6, /*ELONGATE_PRE*/ 3, /*ELONGATE2D_PRE*/ 2,
    /*ARRAY_PRE*/ 9, /*ARRAY2D_PRE*/ 6, /*CIRC_PRE*/ 4, /*LOADP2_SUB*/ 2,
    /*MULR*/ 1, /*SHELL_POST*/ 1, /*ADDR*/ 1, /*ANNULUS*/ 1, /*EXTRUDE_POST*/ 0, /*MAXR_SLOT*/ 0, /*ADDR_SLOT*/ 0,
    /*SAVEP3*/ 0, /*LOADP3*/ 0, /*SAVEP2*/ 0, /*LOADP2*/ 0, /*SAVER*/ 0, /*SETSLOT*/ 1, /*SETR*/ 1,
    /*MIN*/ 0, /*MAX*/ 0, /*DIFF*/ 0, /*XOR*/ 0, /*SUNION*/ 2, /*SDIFF*/ 2, /*SINTER*/ 2,
    /*GATE2D*/ 10, /*GATE3D*/ 12, /*GATEZC*/ 13, /*UBOUND2D*/ 1, /*UBOUND3D*/ 1, /*GATEOB*/ 14, /*CIRC_ORDER*/ 6,
    /*LIP_PUSH*/ 0, /*LIP_POP*/ 0, /*LIP_MUL*/ 1, /*LIP_WRAP*/ 2,
    /*SKIP*/ 3, /*LIP_DOM*/ 5,
};
#define GSDF_GATE_SKIP(w) ((w) & 0x00ffffffu)  // a gate's (and D_SKIP's) last parameter word: words to skip | (number + 1) << 24 (D_SKIP: flags << 24)
#define GSDF_GATE_ID1(w) ((w) >> 24)
#define GSDF_BRICK_MASK_VALID 0x80000000u      // brick mask handed to sdf_eval: bits 0..15 subtree numbers, bit 31 "a centre test produced it"
#define GSDF_SKIP_BIG 1.0e30f
