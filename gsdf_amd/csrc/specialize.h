// specialize.h -- run-time specialisation of the evaluator for one lowered program (hiprtc).
#pragma once
#include <string>
#include <vector>

#include "compile.h"

namespace gsdf_dev {

// Text of gsdf_spec_gen.h: sdf_eval<K, PAIRED> for this program, assembled from the interpreter's own case bodies
// (embedded at build time by gen_embedded.py) in program order with every instruction word and parameter a literal.
std::string spec_source(const Program& p);

// Number of instructions of the program (excluding D_END).
int spec_instruction_count(const Program& p);
// How much of the program depends on the entry x and y alone (instructions flagged D_FLAG_SHXY; an atan2 counts 4, a hypot 1):
// decides whether the column-brick leaf kernel is also built with both passes in one body (kernels_octree.h: BOTH).
int spec_xy_shared_weight(const Program& p);

// Compile kernels.h with the specialised evaluator for `arch` (e.g. "gfx950") and instantiate `name_exprs`
// ("leaf_kernel<4, 4>", "prune_kernel", ...). On success returns true and fills the code object and the lowered
// (mangled) names in the same order; otherwise `log` holds the compiler output. Needs no GPU.
bool spec_compile(const Program& p, const std::string& arch, const std::vector<std::string>& name_exprs,
                  std::vector<char>& code_object, std::vector<std::string>& lowered, std::string& log);

// "hipcc" (out-of-process build with the installed compiler), "hiprtc" (the process's run-time compiler) or "cache": how
// the calling thread's last successful spec_compile got its code object.
const char* spec_last_compiler();
// Key of the last spec_compile of this thread (hash of generated source, device headers, options, compiler identity), and of
// the ahead-of-time kernels of this library (hash of its device sources): what a stored profile must match to describe the code
// that runs now.
const char* spec_last_key();
std::string spec_library_key();

}  // namespace gsdf_dev
