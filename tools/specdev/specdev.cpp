// specdev.cpp -- developer shim around the run-time specialiser (host only, no GPU): builds the kernels of one tree for
// gfx950 with hiprtc and writes the code object to a file, so that register use, scratch and ISA of a specialised build
// can be inspected after every edit of interp.h / kernels.h without rebuilding libgsdfhip.so (tools/specdev.py).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gsdf_program.h"
#include "../../gsdf_amd/csrc/compile.h"
#include "../../gsdf_amd/csrc/specialize.h"

extern "C" int specdev_build(const gsdf_tree* tree, const char* names_semicolon, const char* out_path, char* log_out, size_t log_cap) {
  std::string log;
  int rc = 0;
  try {
    const gsdf_dev::Program pr = gsdf_dev::compile(*tree);
    std::vector<std::string> names;
    std::string cur;
    for (const char* c = names_semicolon; ; c++) {
      if (*c == ';' || *c == 0) { if (!cur.empty()) names.push_back(cur); cur.clear(); if (!*c) break; }
      else cur.push_back(*c);
    }
    std::vector<char> co;
    std::vector<std::string> low;
    if (!gsdf_dev::spec_compile(pr, "gfx950", names, co, low, log)) rc = 2;
    else {
      FILE* f = fopen(out_path, "wb");
      if (!f || fwrite(co.data(), 1, co.size(), f) != co.size()) rc = 3;
      if (f) fclose(f);
    }
  } catch (const std::exception& e) {
    log = e.what();
    rc = 1;
  }
  if (log_out && log_cap) { strncpy(log_out, log.c_str(), log_cap - 1); log_out[log_cap - 1] = 0; }
  return rc;
}
extern "C" int specdev_source(const gsdf_tree* tree, const char* out_path) {
  try {
    const std::string s = gsdf_dev::spec_source(gsdf_dev::compile(*tree));
    FILE* f = fopen(out_path, "wb");
    if (!f) return 3;
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
    return 0;
  } catch (const std::exception&) { return 1; }
}
