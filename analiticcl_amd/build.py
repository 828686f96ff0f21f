"""Build libanx.so in-tree: `python -m analiticcl_amd.build` (hipcc --offload-arch=gfx950, no GPU needed)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> str:
    csrc = os.path.join(HERE, "csrc")
    args = ["make", "-C", csrc, "-s", "-j8"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    out = os.path.join(HERE, "libanx.so")
    if not os.path.exists(out):
        raise RuntimeError("libanx.so was not produced")
    return out


if __name__ == "__main__":
    print(build())
