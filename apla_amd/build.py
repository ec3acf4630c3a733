"""Build libapla_hip.so (gfx950 only) in-tree with hipcc.  No torch headers, no pybind: a plain C-ABI shared library
that Python binds through ctypes (apla_amd/_lib.py).  ``python -m apla_amd.build`` or ``__graft_entry__.build()``."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libapla_hip.so")          # bf16 operands (default, benchmarked)
OUT_F16 = os.path.join(HERE, "libapla_hip_f16.so")  # same sources with -DAPLA_FP16: IEEE fp16 operands
SOURCES = ["errors.cpp", "gemm_nt.hip", "gemm_pp2.hip", "gemm_w4.hip", "gemm_tp.hip", "gemm_lw.hip", "gemm_small.hip", "layernorm.hip", "attention.hip", "apla_dw.hip", "optim.hip", "misc.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
# Sources whose kernels count their own s_waitcnt (LDS-DMA rings, ds_read_tr pipelines, asm operand loads): a register spill puts scratch
# loads and stores into the same in-order vmcnt queue and silently breaks those counts (round 4: the fused attention backward at
# three waves per SIMD spilled 74 registers and returned wrong gradients).  Their objects are only accepted spill-free.
NO_SPILL = ("gemm_nt.hip", "gemm_pp2.hip", "gemm_w4.hip", "gemm_tp.hip", "gemm_lw.hip", "attention.hip", "apla_dw.hip", "layernorm.hip")


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    deps = [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "gemm_tp_asm.inc"), os.path.join(HERE, "..", "include", "apla_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def audit_gemm_tp(defines=(), verbose: bool = True) -> dict:
    """gemm_tp.hip keeps its 160 accumulators in AGPRs under literal names hipcc's allocator cannot see (the file says why).  That is
    sound only while hipcc itself uses no AGPR and spills nothing in those kernels: compile the device code to assembly and check —
    no scratch, no spill, no accumulator-register access outside the file's own asm statements.  Raises on a violation."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm_tp.s")
        cmd = [HIPCC] + FLAGS + list(defines) + ["--cuda-device-only", "-S", os.path.join(CSRC, "gemm_tp.hip"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        text = open(out).read()
    kernels = 0
    in_asm = False
    bad = []
    for n, line in enumerate(text.splitlines(), 1):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")) and re.search(r"v_accvgpr|[\s,]a\[?\d+", line):
            bad.append((n, line.strip()))
    meta = {k: [int(v) for v in re.findall(rf"\.{k}:\s+(\d+)", text)] for k in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "agpr_count", "vgpr_count")}
    kernels = len(meta["agpr_count"])
    if bad:
        raise RuntimeError(f"gemm_tp.hip: hipcc touches accumulator registers outside the kernel's asm statements ({len(bad)} lines, first: {bad[0]})")
    if any(meta["vgpr_spill_count"]) or any(meta["private_segment_fixed_size"]):
        raise RuntimeError(f"gemm_tp.hip: register spills / scratch in a kernel that must have none: {meta}")
    if kernels == 0 or any(a != 160 for a in meta["agpr_count"]) or any(v > 256 for v in meta["vgpr_count"]):
        raise RuntimeError(f"gemm_tp.hip: unexpected register counts {meta}")
    if verbose:
        print(f"gemm_tp audit: {kernels} kernels, registers {meta['vgpr_count']} (160 of them AGPRs), no spill, no scratch, no foreign accumulator access")
    return meta


def spilled_kernels(remarks: str) -> dict:
    """{kernel: spilled VGPRs} for every kernel of a `-Rpass-analysis=kernel-resource-usage` listing that spills vector registers
    (spilled SGPRs go to VGPR lanes by v_writelane: no memory operation, harmless for the wait counts)."""
    import re
    out, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name and int(m.group(1)) > 0:
            out[name] = out.get(name, 0) + int(m.group(1))
    return out


def build(force: bool = False, verbose: bool = True, fp16: bool = True) -> str:
    """Compile libapla_hip.so and (fp16=True) libapla_hip_f16.so."""
    _build_one(OUT, "build", [], force, verbose)
    if fp16:
        _build_one(OUT_F16, os.path.join("build", "f16"), ["-DAPLA_FP16"], force, verbose)
    return OUT


def _build_one(out, objsub, defines, force, verbose):
    objdir = os.path.join(HERE, objsub)
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, src):
            cmd = [HIPCC] + FLAGS + defines + (["-x", "hip"] if s.endswith(".cpp") else []) + \
                  (["-Rpass-analysis=kernel-resource-usage"] if s in NO_SPILL else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if "-Rpass-analysis=kernel-resource-usage" in cmd:
            spilled = spilled_kernels(r.stderr)
            if spilled:
                os.remove(cmd[-1])      # never link an object whose hand-counted waits cannot be trusted
                raise RuntimeError(f"{os.path.basename(cmd[-3])}: register spills in kernels that count their own waits: {spilled}")
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if any("gemm_tp.hip" in " ".join(j) for j in jobs):
        audit_gemm_tp(defines, verbose)     # a fresh gemm_tp object is only accepted with its register audit
    objs = [os.path.join(objdir, s.rsplit(".", 1)[0] + ".o") for s in SOURCES]
    if jobs or not os.path.exists(out):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
