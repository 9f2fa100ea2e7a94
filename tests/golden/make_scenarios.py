"""Regenerates cityflow_amd/data/scenarios/* from the reference tree (only runnable where /root/reference exists).

  example_1x1  : the reference's own examples/{roadnet,flow}.json (the only scenario its tests use)
  grid_6x6     : tools/generator/generate_grid_scenario.py 6 6 --tlPlan --interval 1.0   
  grid_30x30   : tools/generator/generate_grid_scenario.py 30 30 --tlPlan --interval 1.0

Outputs are stored gzip-compressed (mtime 0, so the bytes are reproducible); cityflow_amd.scenarios
materialises them into a work directory together with a config.json at test / bench time.
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

REF = os.environ.get("CITYFLOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "cityflow_amd", "data", "scenarios")


def pack(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(src, "rb") as f, open(dst, "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", fileobj=raw, compresslevel=9, mtime=0) as g:
            shutil.copyfileobj(f, g)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at " + REF)
    for name in ("roadnet", "flow"):
        pack(os.path.join(REF, "examples", name + ".json"), os.path.join(OUT, "example_1x1", name + ".json.gz"))
    gen = os.path.join(REF, "tools", "generator", "generate_grid_scenario.py")
    for n in (6, 30):
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.check_call([sys.executable, gen, str(n), str(n), "--tlPlan", "--interval", "1.0", "--dir", tmp,
                                   "--roadnetFile", "roadnet.json", "--flowFile", "flow.json"],
                                  cwd=os.path.dirname(gen))
            for name in ("roadnet", "flow"):
                pack(os.path.join(tmp, name + ".json"), os.path.join(OUT, "grid_%dx%d" % (n, n), name + ".json.gz"))
    print("scenarios written to", OUT)


if __name__ == "__main__":
    main()
