#!/usr/bin/env python3
"""Register / scratch / code-size table of the hot kernels, from the compiler's own resource remarks -- the guard the
11-parameter gather-GEMM template needs: adding a mode to csrc/gather_gemm.hip must not move the register allocation or the
code size of the fp32 instances that carry the decode (round 3 lost 8 % to exactly that: 171 -> 184 VGPRs from one run-time branch).

    python tools/kernel_resources.py            # recompile the listed sources for gfx950, print the table
    python tools/kernel_resources.py --write    # ... and rewrite profiles/kernel_resources.json (commit it with the change)
    python tools/kernel_resources.py --check    # exit 1 if an instance differs from the committed table (tests/test_abi_and_host.py)

No GPU needed (hipcc cross-compiles); the four sources take about three minutes."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "shallow-ntc_amd" / "csrc"
TABLE = ROOT / "profiles" / "kernel_resources.json"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only",
         "-c", "-o", "/dev/null"]
# source -> the instances (demangled-name fragments) whose numbers are pinned
WATCH = {
    "gg_inst_vec.hip": {
        "gg 128x128 stream-K (HS3 twin, column-major)": "gg_kernelILi2ELi2ELi2ELi2ELb1ELb0ELb0ELb0ELi0ELb0ELb1EE",
        "gg 128x128 (strip-major)": "gg_kernelILi2ELi2ELi2ELi2ELb1ELb0ELb0ELb0ELi0ELb0ELb0EE",
        "gg 128x96 (HS2, 5x5/2 layers)": "gg_kernelILi1ELi3ELi4ELi1ELb1ELb0ELb0ELb0ELi0ELb0ELb0EE",
        "gg 128x64": "gg_kernelILi1ELi2ELi4ELi1ELb1ELb0ELb0ELb0ELi0ELb0ELb0EE",
        "gg 64x64": "gg_kernelILi1ELi1ELi2ELi2ELb1ELb0ELb0ELb0ELi0ELb0ELb0EE",
    },
    "gg_inst_fuse.hip": {"gg fused ResidualBlock tail": "gg_kernelILi1ELi3ELi4ELi1ELb1ELb0ELb0ELb0ELi0ELb1ELb0EE"},
    "rb_fused.hip": {"rb_kernel<192>": "rb_kernelILi192EE"},
    "syn_fused.hip": {"syn_kernel<24, true>": "syn_kernelILi24ELb1EE", "syn_kernel<12, false>": "syn_kernelILi12ELb0EE",
                      "syn_kernel<24, false>": "syn_kernelILi24ELb0EE"},
    "rgb_conv.hip": {"rgb_conv_kernel<6, 5, false> (ELIC first layer)": "rgb_conv_kernelILi6ELi5ELb0EE",
                     "rgb_conv_kernel<8, 5, true>": "rgb_conv_kernelILi8ELi5ELb1EE"},
    "up_small.hip": {"up_small_kernel<5, 2, 3>": "up_small_kernelILi5ELi2ELi3EE", "up_small_kernel<9, 4, 3>": "up_small_kernelILi9ELi4ELi3EE"},
    "entropy.hip": {"scale_normal_kernel<false>": "scale_normal_kernelILb0EE", "factorized_fast_kernel<3, 3>": "factorized_fast_kernelILi3ELi3EE"},
}
FIELDS = {"VGPRs": r"\bVGPRs: (\d+)", "AGPRs": r"AGPRs: (\d+)", "SGPRs": r"TotalSGPRs: (\d+)", "scratch_bytes": r"ScratchSize \[bytes/lane\]: (\d+)",
          "waves_per_simd": r"Occupancy \[waves/SIMD\]: (\d+)", "vgpr_spills": r"VGPRs Spill: (\d+)"}


def measure():
    out = {}
    for src, inst in WATCH.items():
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(CSRC / src)], capture_output=True, text=True, cwd=str(CSRC))
        if r.returncode != 0:
            raise SystemExit(f"{src}: compile failed\n{r.stderr[-2000:]}")
        blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
        for label, frag in inst.items():
            hit = [b for b in blocks if frag in b.split()[0]]
            if len(hit) != 1:
                raise SystemExit(f"{src}: {len(hit)} kernels match {frag!r}")
            out[label] = {k: int(re.search(p, hit[0]).group(1)) for k, p in FIELDS.items() if re.search(p, hit[0])}
    return out


def main():
    now = measure()
    for k, v in now.items():
        print(f"{k:48s} " + "  ".join(f"{a}={b}" for a, b in v.items()))
    if "--write" in sys.argv:
        TABLE.write_text(json.dumps(now, indent=1) + "\n")
        print("wrote", TABLE.relative_to(ROOT))
        return 0
    if "--check" in sys.argv:
        want = json.loads(TABLE.read_text())
        bad = {k: (want.get(k), v) for k, v in now.items() if want.get(k) != v}
        if bad:
            for k, (w, g) in bad.items():
                print(f"CHANGED {k}: committed {w}, now {g}")
            print("if intended: python tools/kernel_resources.py --write, and re-measure the layer tables")
            return 1
        print("matches", TABLE.relative_to(ROOT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
