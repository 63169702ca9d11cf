// compile.h -- host lowering of a gsdf tree blob to the device instruction stream.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/gsdf_program.h"

namespace gsdf_dev {

struct Program {
  std::vector<uint32_t> code;  // dev_ops.h stream, terminated by D_END
  int nslots = 0;              // LDS scratch slots per lane
  int lip_depth = 0;           // slots of the interval stack (dev_ops.h: D_LIP_PUSH / _POP); prune_kernel's columns = nslots + lip_depth
  int n_skip_ids = 0;          // operand subtrees that carry a brick-mask number (dev_ops.h: D_SKIP); 0: the leaf kernels get no masks
  bool is2d = false;           // root takes 2D positions
  float bb[6] = {0};
  // Set when the whole field is known to be >= the Euclidean distance to `exact_bb` (exact-distance primitives under
  // translation / union / difference, see compile.cpp: exact_box): lets a renderer decide "farther than r from the
  // surface" for lattice points outside that box by more than r without evaluating them.
  bool has_exact_bb = false;
  float exact_bb[6] = {0};
};

// RN(1/d) for the device's exact division by a wave-uniform divisor, 0 if d is not eligible.
float recip_for(float d);

// Lower-bound region of node `node`'s subtree (compile.cpp: lower_region), for tests: kind 0 = no claim, 1 = box
// {min xyz, max xyz}, 2 = z-cylinder {cx cy r z0 z1 rs rin}. Throws on malformed trees.
int region_of(const gsdf_tree& t, uint32_t node, float params[8]);

// Throws std::runtime_error on malformed trees. max_code_words bounds unrolling blow-up.
Program compile(const gsdf_tree& t, size_t max_code_words = (1u << 22));

}  // namespace gsdf_dev
