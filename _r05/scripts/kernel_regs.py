"""VGPR / spill / LDS numbers of the gfx950 kernels in one object file (no GPU needed).
   python3 scripts/kernel_regs.py wcmc_amd/csrc/conv_bf16x3.o [name filter]"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    local = os.path.join(d, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", local], check=True, capture_output=True, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f][0]
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(d, co)], check=True, capture_output=True, text=True).stdout
rows = []
for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, e).group(1)
    rows.append((g("name"), int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("sgpr_spill_count")), int(g("group_segment_fixed_size")),
                 int(e.split("\n")[0].strip() or 0)))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("wcmc::", "").replace("(wcmc::XWRowsParams)", "").replace("(wcmc::XIgemmParams)", "").replace("(wcmc::XWgradParams)", "")
    if flt in n:
        print("%-78s vgpr %3d agpr %3d  vgpr spill %3d  sgpr spill %3d  static lds %6d" % (n[:78], r[1], r[5], r[2], r[3], r[4]))
