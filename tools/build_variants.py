"""Developer tool (run HERE, where hipcc cross-compiles; the GPU box then only measures): differently built device libraries for
a same-state A/B on the GPU, and what the compiler made of the kernels in each.
usage: python tools/build_variants.py NAME="-DCFX_CROSS2_WAVES=6 -DCFX_CROSS2_WORK=1792" NAME2="..." [--kernels k_cross2,kd_action]
  -> gpurun_exp/lib_NAME.so ... (git-ignored, shipped by gpurun) and a table of registers / spills / LDS / scratch per kernel
then, about a minute of GPU time for several variants at once:
  gpurun -- 'python tools/exp_big.py gpurun_exp/lib_NAME.so gpurun_exp/lib_NAME2.so'        (100x100, dense layout)
  gpurun -- 'python tools/exp_bench.py gpurun_exp/lib_NAME.so ...'                            (30x30, ring layout)
  gpurun -- 'python tools/ab_bench.py gen_100x100 rounds=2 layout=dense layout=dense,lib=gpurun_exp/lib_NAME.so'   (interleaved)
The knobs are listed in DESIGN.md section 9, item 0."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "cityflow_amd", "csrc", "hip")
OUT = os.path.join(ROOT, "gpurun_exp")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-Wall",
        "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-I" + HIP]


def build(name, flags):
    src = os.path.join(HIP, "cfx_hip.hip")
    lib = os.path.join(OUT, "lib_%s.so" % name)
    asm = os.path.join(OUT, "lib_%s.s" % name)
    subprocess.check_call(BASE + flags + ["-shared", src, "-o", lib])
    subprocess.check_call(BASE + flags + ["--cuda-device-only", "-S", src, "-o", asm], stderr=subprocess.DEVNULL)
    meta = {}
    for block in open(asm).read().split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", block).group(1)
        short = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
        meta[short] = (int(g("vgpr_count")), int(g("vgpr_spill_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")))
    os.remove(asm)
    return name, lib, meta


def main():
    os.makedirs(OUT, exist_ok=True)
    want = None
    variants = [("default", [])]
    for a in sys.argv[1:]:
        if a.startswith("--kernels"):
            want = a.split("=", 1)[1].split(",")
        else:
            name, flags = a.split("=", 1)
            variants.append((name, flags.split()))
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda v: build(*v), variants))
    kernels = sorted(results[0][2])
    print("%-58s" % "kernel (vgpr / spilled / lds / scratch)" + "".join("%-24s" % r[0] for r in results))
    for k in kernels:
        if want and not any(w in k for w in want):
            continue
        rows = [r[2].get(k) for r in results]
        if not want and all(x == rows[0] for x in rows):
            continue  # (only what a variant changed)
        print("%-58s" % k[:57] + "".join("%-24s" % ("%d / %d / %d / %d" % x if x else "-") for x in rows))
    for r in results[1:]:
        print("built", os.path.relpath(r[1], ROOT))


if __name__ == "__main__":
    main()
