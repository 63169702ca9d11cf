import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
sys.argv = ["disasm.py"]
src = open("/root/repo/tools/disasm.py").read()
from scaffold.builder import Builder
bld = Builder()
ttf = open("/root/repo/tests/golden/iso-3098.ttf", "rb").read()
t2 = bld.TextLine(ttf, "gsdf MI355X")
tb = t2.Bounds()
w, h = float(tb[3] - tb[0]), float(tb[4] - tb[1])
plate = bld.Translate(bld.NewBox(w + 0.3, h + 0.3, 0.06, 0.01), float(tb[0] + tb[3]) / 2, float(tb[1] + tb[4]) / 2, -0.08)
sh = bld.Union(bld.Extrude(t2, 0.12), plate)
print(tb, sh.Bounds())
src = src.replace('sh = Builder().Scene(sys.argv[1] if len(sys.argv) > 1 else "npt-flange")', "pass")
exec(compile(src, "/root/repo/tools/disasm.py", "exec"), {"__file__": "/root/repo/tools/disasm.py", "__name__": "__main__", "sh": sh})
