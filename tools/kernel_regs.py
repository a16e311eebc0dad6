"""Register / scratch / LDS use of the shipped gfx950 kernels (no GPU needed): reads the AMDGPU metadata notes of the code
objects inside libmapnet_hip.so.  usage: python tools/kernel_regs.py [kernel-name-regex]"""
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_audit import ROOT, code_objects  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
lib = sys.argv[2] if len(sys.argv) > 2 else ROOT + "/geomapnet_amd/libmapnet_hip.so"
for co in code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(co)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if not pat.search(name):
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("vgpr %3d agpr %3d spill %3d sgpr %3d lds %6d scratch %4d  %s" % (g("vgpr_count"), int(blk.split()[0]), g("vgpr_spill_count"),
              g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), dem[:150]))
