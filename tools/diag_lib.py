"""The diagnostic (L2_STAMPS) build of the library: never shipped and never pushed to a GPU box, so the tools that read in-kernel
clock stamps build it where they run (`make -C llama2.ts_amd/csrc stamps` -> gpurun_out/diag/libllama2hip_stamps.so)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "diag", "libllama2hip_stamps.so")


def use():
    """Build the stamps library if it is missing or older than the sources and point the runtime at it (before it is imported)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "llama2.ts_amd", "csrc"), "stamps"], check=True, stdout=subprocess.DEVNULL)
    os.environ["L2_LIB_PATH"] = PATH
    return PATH
