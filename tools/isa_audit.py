"""Static audit of the shipped gfx950 code objects (no GPU needed): carves them out of libmapnet_hip.so's offload
bundles, disassembles them with llvm-objdump and reports, per kernel that contains MFMAs, the order of barriers, vmcnt waits, LDS-DMA issues, LDS reads, MFMAs and
scratch accesses around its K loop.  A scratch
reload or a vmcnt(0) between the loop's barrier and its MFMAs is a stall behind the LDS-DMA queue.
usage: python tools/isa_audit.py [kernel-name-regex]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        b = m.start()
        num = struct.unpack_from("<Q", data, b + 24)[0]
        off = b + 32
        for _ in range(num):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple:
                yield data[b + o:b + o + sz]


def signatures(pattern, out=None):
    """{kernel name: (instructions, MFMAs, event string)} for the kernels whose mangled name matches `pattern`"""
    pat = re.compile(pattern)
    lib = os.path.join(ROOT, "geomapnet_amd", "libmapnet_hip.so")
    seen = set()
    result = {}
    for k, co in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout.split("\n")
        starts = [(i, m.group(1)) for i, l in enumerate(txt) for m in [re.match(r"^[0-9a-f]+ <(.+)>:$", l)] if m]
        for n, (i, name) in enumerate(starts):
            if not pat.search(name) or name in seen:
                continue
            body = txt[i:starts[n + 1][0] if n + 1 < len(starts) else len(txt)]
            mf = [j for j, l in enumerate(body) if "v_mfma" in l]
            if not mf:
                continue
            seen.add(name)
            # event string of the code around the MFMAs (the K loop with the head that precedes it in the binary):
            # B barrier, V<n> s_waitcnt vmcnt(n), L<n> s_waitcnt lgkmcnt(n) alone, D LDS-DMA, R ds_read, M MFMA, S scratch,
            # J branch; runs are counted
            seg = body[max(0, mf[0] - 160):mf[-1] + 80]
            ev = []
            for l in seg:
                t = l.split("//")[0].strip()
                e = None
                if "v_mfma" in t:
                    e = "M"
                elif t.startswith("s_barrier"):
                    e = "B"
                elif t.startswith("s_waitcnt"):
                    m = re.search(r"vmcnt\((\d+)\)", t)
                    e = "V%s" % m.group(1) if m else None
                elif t.startswith("buffer_load") and t.endswith("lds") or t.startswith("global_load_lds"):
                    e = "D"
                elif t.startswith("ds_read"):
                    e = "R"
                elif t.startswith("scratch_"):
                    e = "S!"
                elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                    e = "J"
                if e is None:
                    continue
                if ev and ev[-1][0] == e:
                    ev[-1][1] += 1
                else:
                    ev.append([e, 1])
            sig = " ".join(e if n == 1 else "%s%s%d" % (e, "x", n) for e, n in ev)
            result[name] = (len(body), len(mf), sig)
    return result


def main():
    for name, (n, m, sig) in signatures(sys.argv[1] if len(sys.argv) > 1 else "igemm_kernelIDF16|wgrad_dma_kernel|conv_halo").items():
        print("%s\n    %d instructions, %d MFMAs\n    %s" % (name[:110], n, m, sig))

if __name__ == "__main__":
    main()
