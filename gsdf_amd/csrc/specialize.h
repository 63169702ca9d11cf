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

// Compile kernels.h with the specialised evaluator for `arch` (e.g. "gfx950") and instantiate `name_exprs`
// ("leaf_kernel<4, 4>", "prune_kernel", ...). On success returns true and fills the code object and the lowered
// (mangled) names in the same order; otherwise `log` holds the compiler output. Needs no GPU.
bool spec_compile(const Program& p, const std::string& arch, const std::vector<std::string>& name_exprs,
                  std::vector<char>& code_object, std::vector<std::string>& lowered, std::string& log);

// "hipcc" (out-of-process build with the installed compiler), "hiprtc" (the process's run-time compiler) or "cache": how
// the calling thread's last successful spec_compile got its code object.
const char* spec_last_compiler();

}  // namespace gsdf_dev
