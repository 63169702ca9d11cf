#!/usr/bin/env python3
"""Build container only: the node struct DECLARATIONS of the reference (names, field names, field types -- no code) that the
Go flattener under integration/go copies fields from -> tests/golden/reference_node_structs.json. tests/test_integration_lock.py
checks the flattener against this fixture everywhere, and the fixture against /root/reference where that exists."""
import json
import os
import re
import sys

FILES = ["primitives.go", "primitives2d.go", "operations.go", "operations2d.go", "forge/threads/threads.go"]


def structs(ref="/root/reference"):
    out = {}
    for f in FILES:
        src = open(os.path.join(ref, f)).read()
        pkg = re.search(r"^package (\w+)", src, re.M).group(1)
        for m in re.finditer(r"^type (\w+) struct \{(.*?)^\}", src, re.S | re.M):
            body = re.sub(r"//[^\n]*", "", m.group(2))
            fields, embedded = [], []
            for line in body.split("\n"):
                line = line.strip()
                if not line:
                    continue
                parts = line.split()
                if len(parts) == 1:          # embedded struct
                    embedded.append(parts[0])
                    continue
                typ = parts[-1]
                for name in " ".join(parts[:-1]).split(","):
                    fields.append([name.strip(), typ])
            # a node type is a shader: it writes a GLSL body (and has an Evaluate in cpu_evaluators.go, own or promoted)
            is_node = re.search(r"^func \(\w+ \*%s\) AppendShaderBody\(" % m.group(1), src, re.M) is not None
            out[m.group(1)] = {"file": f, "line": src[:m.start()].count("\n") + 1, "package": pkg, "fields": fields, "embedded": embedded, "node": is_node}
    return out


if __name__ == "__main__":
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_node_structs.json")
    json.dump(structs(*sys.argv[1:]), open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)
