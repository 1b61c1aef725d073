"""Register / LDS / scratch use of every kernel of the library, from the compiler's own metadata (no GPU needed):
     python tools/kernel_resources.py [out.txt]
Compiles each tf2_gnn_amd/csrc/*.hip to gfx950 assembly with the flags of tf2_gnn_amd/build.py (hipcc -S --cuda-device-only)
and reads the amdhsa metadata of every kernel: VGPRs (arch + accumulation), SGPRs, spilled registers, scratch bytes, static
LDS bytes, and the waves per SIMD the register count allows (512 unified registers per lane and SIMD on gfx950, granule 8).
A kernel that spills, or whose occupancy drops after an edit, shows up here before it shows up in a profile."""
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "tf2_gnn_amd" / "csrc"


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return list(names)


def one(src: Path):
    with tempfile.TemporaryDirectory() as d:
        asm = Path(d) / (src.stem + ".s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", f"-I{ROOT / 'include'}", f"-I{CSRC}",
               "--cuda-device-only", "-S", str(src), "-o", str(asm)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            return src.name, [], res.stderr[-400:]
        text = asm.read_text()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", text, re.S):
        body = m.group(0)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, body) else 0
        name = re.search(r"\.name:\s+(\S+)", body).group(1)
        rows.append(dict(name=name, agpr=get("agpr_count"), vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), spill=get("vgpr_spill_count") + get("sgpr_spill_count"),
                         scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"), maxwg=get("max_flat_workgroup_size")))
    return src.name, rows, ""


def main():
    srcs = sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(one, srcs))
    lines = ["kernel resources from hipcc -S metadata (gfx950; vgpr = arch + accumulation registers; waves/SIMD = floor(512 / ceil8(vgpr)), max 8)", ""]
    total = spilling = 0
    for fname, rows, err in results:
        lines.append(f"== {fname}" + (f"   COMPILE ERROR: {err}" if err else f"   ({len(rows)} kernels)"))
        names = demangle([r["name"] for r in rows])
        for r, n in sorted(zip(rows, names), key=lambda t: -t[0]["vgpr"]):
            total += 1
            spilling += 1 if (r["spill"] or r["scratch"]) else 0
            v = max(r["vgpr"], 1)
            waves = min(8, 512 // (((v + 7) // 8) * 8))
            n = re.sub(r"\(.*", "", n)
            lines.append(f"  {n[:96]:96s} vgpr {r['vgpr']:4d} (acc {r['agpr']:3d}) sgpr {r['sgpr']:3d} spill {r['spill']:3d} scratch {r['scratch']:5d} B"
                         f" static LDS {r['lds']:6d} B  threads {r['maxwg']:4d}  waves/SIMD {waves}")
    lines.insert(1, f"{total} kernels, {spilling} with spills or scratch")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(out)
    print(out if len(sys.argv) <= 1 else f"wrote {sys.argv[1]}: {total} kernels, {spilling} with spills or scratch")


if __name__ == "__main__":
    main()
