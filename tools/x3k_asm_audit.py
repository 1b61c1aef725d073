"""Audit of the streaming kernel's hand-placed LDS reads (csrc/gemm_x3.hip, gemm_x3k_kernel).

The B pieces are requested by `asm volatile` ds_read_b128 statements one group of MFMAs ahead of their use; the compiler does
not know that the destination registers of such a read are still in flight.  Inside the step code (VALU + MFMA) that is
harmless; a spill, a branch or an LDS write placed between a request and the `s_waitcnt lgkmcnt(0)` that ends a tile's chain is
not (a spill saves the OLD register contents - the one bug this kernel had).  This script compiles gemm_x3.hip for gfx950 with
-save-temps and checks every instantiation:   python tools/x3k_asm_audit.py        (exit code 1 on a finding; ~70 s)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", f"-I{ROOT}/include",
               f"-I{ROOT}/tf2_gnn_amd/csrc", "-c", f"{ROOT}/tf2_gnn_amd/csrc/gemm_x3.hip", "-o", f"{tmp}/gemm_x3.o", "-save-temps=obj"]
        subprocess.run(cmd, check=True, cwd=tmp)
        asm = open(f"{tmp}/gemm_x3-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    names = re.findall(r"^(_ZN5tfgnn15gemm_x3k_kernel\w+): ", asm, re.M)
    findings = 0
    for name in names:
        i0 = asm.index("\n" + name + ": ")
        lines = asm[i0 : asm.index("s_endpgm", i0)].split("\n")
        pending, bad = False, []
        for i, line in enumerate(lines):
            t = line.strip()
            in_asm = lines[i - 1].strip().startswith(";;#ASMSTART")
            if in_asm and t.startswith("ds_read_b128"):
                pending = True
            if in_asm and t.startswith("s_waitcnt lgkmcnt(0)"):
                pending = False
            if pending and (t.startswith("scratch_store") or t.startswith("s_cbranch") or t.startswith("ds_write") or t.startswith("s_barrier")):
                bad.append((i, t[:60]))
        print(f"{name}: {'OK' if not bad else 'FINDINGS ' + str(bad[:4])}")
        findings += len(bad)
    print(f"{len(names)} instantiations, {findings} finding(s)")
    return 1 if findings or not names else 0


if __name__ == "__main__":
    sys.exit(main())
