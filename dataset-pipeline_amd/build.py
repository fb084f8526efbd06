"""Build libe3dhip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python dataset-pipeline_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the reference is built without FMA
contraction (CMakeLists.txt:82) and correspondence counts are only identical if the f32 arithmetic is.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SOURCES = ["e3d_icp.hip", "e3d_comm.hip", "e3d_icp_kernels.hip", "e3d_sort.hip", "e3d_normals.hip", "e3d_reg.hip", "e3d_multires.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def lib_path():
    return os.path.join(LIBDIR, "libe3dhip.so")


# per-file extras.  No SLP vectorisation for the ICP and registration kernels: the compiler otherwise packs adjacent scalar f32
# operations of their long expression trees into v_pk_*_f32 and spends more on moves than it saves (measured on one MI355X:
# k_lm_cost_multi 2.00 -> 1.67 ms, k_lm_pass<3> 0.97 -> 0.91 ms, k_reg_pass1 0.503 -> 0.447 ms; tools/micro/lm_variants.hip, DESIGN
# 4.2).  The kNN kernels of e3d_normals.hip gain from it (1.72 vs 1.34 G normals/s) and keep it.  Rounding is the same either way.
EXTRA_FLAGS = {"e3d_icp_kernels.hip": ["-fno-slp-vectorize"], "e3d_reg.hip": ["-fno-slp-vectorize"]}


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    inc = os.path.join(HERE, "..", "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(s, []) + os.environ.get("E3D_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]
        # the flags are part of the object's identity (E3D_EXTRA_HIPCC_FLAGS=-DE3D_NT=0 and the like are used for A/B timing): the
        # command line is kept next to the object and a different one rebuilds it
        stamp = obj + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or _newer(src, obj) or any(_newer(hd, obj) for hd in hdrs):
            if verbose:
                print(" ".join(cmd))
            if os.path.exists(stamp):
                os.remove(stamp)
            procs.append((s, subprocess.Popen(cmd), stamp, " ".join(cmd)))
    failed = []
    for s, p, stamp, line in procs:
        if p.wait() != 0:
            failed.append(s)
        else:
            with open(stamp, "w") as f:
                f.write(line)
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed))
    so = lib_path()
    if force or procs or not os.path.exists(so) or any(_newer(o, so) for o in objs):   # objects built by hand count too
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
