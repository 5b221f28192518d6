#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in libspindyn.so, read from the gfx950 CODE OBJECT (the AMDGPU metadata note),
not from rocprofv3's trace columns (which print VGPR granules-in-use of the dispatch packet and 0 for dynamic LDS:
VERDICT r03, evidence hygiene 9).  Usage: python3 profiles/codeobj_meta.py [lib.so] -> one line per kernel; importable: meta(path)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("SD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "spindynamics.jl_amd", "libspindyn.so")


def meta(path=LIB):
    """{demangled kernel name: {vgpr, agpr, sgpr, lds_static, scratch, max_wg}}"""
    notes = ""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        blob = open(fat, "rb").read()
        # one offload bundle per translation unit, concatenated in the section
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for k in range(len(starts) - 1):
            part, co = os.path.join(td, "part%d.bin" % k), os.path.join(td, "k%d.co" % k)
            open(part, "wb").write(blob[starts[k]:starts[k + 1]])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout + "\n"
    out, cur = {}, {}
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds_static",
            ".private_segment_fixed_size": "scratch", ".max_flat_workgroup_size": "max_wg"}

    def flush():
        if "name" in cur:
            out[cur.pop("name")] = dict(cur)
        cur.clear()

    for line in notes.splitlines():
        m = re.match(r"\s*(-\s+)?(\.[a-z_]+):\s+(.*)$", line)
        if not m:
            continue
        if m.group(1) and cur:
            flush()
        k, v = m.group(2), m.group(3).strip()
        if k == ".name":
            cur["name"] = v.strip("'\"")
        elif k in keys:
            try:
                cur[keys[k]] = int(v)
            except ValueError:
                pass
    flush()
    names = list(out)
    if names:
        import shutil
        filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.join(LLVM, "llvm-cxxfilt")
        try:
            dem = subprocess.run([filt], input="\n".join(names), text=True, capture_output=True).stdout.splitlines()
            out = {d.strip(): out[n] for n, d in zip(names, dem)}
        except OSError:
            pass                      # no demangler: mangled names
    return out


def waves_per_simd(vgpr, agpr=0):
    """gfx950: 512 VGPRs per SIMD lane (unified VGPR + AGPR file), allocation granule 8, at most 8 waves."""
    tot = max(1, ((vgpr + agpr + 7) // 8) * 8)
    return min(8, 512 // tot)


if __name__ == "__main__":
    for n, d in sorted(meta(sys.argv[1] if len(sys.argv) > 1 else LIB).items()):
        print("%-150s vgpr=%-3d agpr=%-3d sgpr=%-3d lds_static=%-6d scratch=%-4d waves/SIMD<=%d" % (
            n[:150], d.get("vgpr", -1), d.get("agpr", 0), d.get("sgpr", -1), d.get("lds_static", 0), d.get("scratch", 0),
            waves_per_simd(d.get("vgpr", 512), d.get("agpr", 0))))
