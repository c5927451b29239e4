"""stt_amd/build.py -- builds stt_amd/lib/libstt.so (HIP kernels + C-ABI) for gfx950 with hipcc.

In-tree build, no JIT cache: the .so travels with the repository snapshot to the GPU box.
    python -m stt_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libstt.so")
# The same objects plus the test hooks of include/stt_amd_test.h (STTX_Test*, STTX_Debug*, the timing-probe kernels): what tests/ load.  The
# shipped libstt.so carries none of them.  Only the sources below know about STT_TEST_HOOKS and are compiled twice.
LIB_TEST = os.path.join(HERE, "lib", "libstt_test.so")
HOOK_SOURCES = ["kernels_am.hip", "kernels_i8.hip", "ctc.hip", "api.cpp", "fleet.cpp"]
SOURCES = ["kernels_am.hip", "kernels_i8.hip", "ctc.hip", "hostutil.cpp", "scorer_dev.cpp", "model.cpp", "tflite_reader.cpp", "engine.cpp", "api.cpp", "fleet.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-x", "hip", "-Wno-unused-result"]
# ctc.hip: the search kernels run 1024 threads per workgroup (128 registers per lane) through one very long timestep loop.  Machine-level
# loop-invariant code motion hoists every `thread index * 4 + LDS constant` address and every f64 polynomial coefficient of the bit-exact
# logf / expf out of that loop and then SPILLS them (scratch loads where one v_add / two v_mov would do): 41 spilled vector registers
# with it, none without (benchmarks/kernel_resources.sh).
EXTRA_FLAGS = {"ctc.hip": ["-mllvm", "-disable-machine-licm"]}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", f) for f in ("coqui-stt.h", "stt_amd.h", "stt_amd_test.h")]
    hdr_mtime = max(os.path.getmtime(h) for h in headers)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        if force or _newer(s, o) or hdr_mtime > os.path.getmtime(o):
            jobs.append((s, o, []))
        if src in HOOK_SOURCES:
            oh = os.path.join(OBJ, src.rsplit(".", 1)[0] + "_hooks.o")
            if force or _newer(s, oh) or hdr_mtime > os.path.getmtime(oh):
                jobs.append((s, oh, ["-DSTT_TEST_HOOKS"]))

    def cc(job):
        s, o, defs = job
        cmd = [hipcc] + FLAGS + defs + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    for lib, suffix in ((LIB, {}), (LIB_TEST, {src: "_hooks" for src in HOOK_SOURCES})):
        objs = [os.path.join(OBJ, src.rsplit(".", 1)[0] + suffix.get(src, "") + ".o") for src in SOURCES]
        if jobs or not os.path.exists(lib) or _newer(os.path.join(CSRC, "libstt.map"), lib):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread", "-Wl,--version-script=" + os.path.join(CSRC, "libstt.map")]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    build_tools(force=force, verbose=verbose)
    return LIB


TOOLS_SRC = os.path.join(HERE, "tools", "scorer_tools.cpp")
TOOLS_BIN = os.path.join(HERE, "lib", "stt_scorer_tools")
CLIENT_SRC = os.path.join(HERE, "tools", "stt_client.cpp")
CLIENT_BIN = os.path.join(HERE, "lib", "stt")


def build_tools(force=False, verbose=True):
    """Host-only scorer packaging tool (generate_scorer_package restated + synthetic LM writer)."""
    if force or _newer(TOOLS_SRC, TOOLS_BIN):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-o", TOOLS_BIN, TOOLS_SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # the `stt` command-line client: plain C++ on include/coqui-stt.h + include/stt_amd.h, linked against libstt.so
    if os.path.exists(LIB) and (force or _newer(CLIENT_SRC, CLIENT_BIN) or _newer(LIB, CLIENT_BIN)):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-o", CLIENT_BIN, CLIENT_SRC, "-L" + os.path.dirname(LIB), "-lstt",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return TOOLS_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
