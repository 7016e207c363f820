"""ctypes binding of libpcc_hip.so (the C-ABI declared in include/pcc_hip.h).

The library is built in-tree by ``build()`` (``make -C csrc``; hipcc --offload-arch=gfx950) and
must be present for any operator call: there is no CPU fallback and no second backend.  A
missing or unloadable library raises ``RuntimeError`` at first use.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libpcc_hip.so")
_lib = None

c_void_p, c_int, c_i32, c_i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64

# name -> (restype, argtypes).  Kept in lock-step with include/pcc_hip.h (tests check every
# symbol declared there is exported and listed here).
SIGNATURES = {
    "pcc_version": (c_int, []),
    "pcc_last_error": (ctypes.c_char_p, []),
    "pcc_device_count": (c_int, []),
    "pcc_device_name": (c_int, [c_int, ctypes.c_char_p, c_int]),
    "pcc_hash_capacity": (c_i64, [c_i64]),
    "pcc_hash_build": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "pcc_hash_lookup": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_i64, c_void_p, c_void_p]),
    "pcc_scan_scratch_elems": (c_i64, [c_i64]),
    "pcc_stride_map": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_children": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_kernel_map": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_pair_count": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "pcc_conv_packed_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "pcc_conv_pack_weights": (c_int, [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "pcc_order_scratch_bytes": (c_i64, [c_i64]),
    "pcc_order_rows_by_mask": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_permute_map_rows": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "pcc_conv_fwd": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32,
                             c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_conv_small_max": (c_i64, [c_i64]),
    "pcc_small_map_max": (c_i64, []),
    "pcc_small_paths": (c_i32, [c_i32]),
    "pcc_small_kernel_map": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_conv_packed_elems_bf16": (c_i64, [c_i32, c_i32, c_i32]),
    "pcc_conv_pack_weights_bf16": (c_int, [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "pcc_conv_fwd_bf16": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_i64,
                                  c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_conv_packed_elems_x3": (c_i64, [c_i32, c_i32, c_i32]),
    "pcc_conv_pack_weights_x3": (c_int, [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "pcc_conv_fwd_x3": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_i64,
                                c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_epilogue_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p]),
    "pcc_cast_colsum_scratch_elems": (c_i64, [c_i32]),
    "pcc_cast_colsum": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_epilogue_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_im2col_thin": (c_int, [c_void_p, c_i32, c_void_p, c_i64, c_i32, c_void_p, c_i32, c_void_p]),
    "pcc_gather_sum_fwd": (c_int, [c_void_p, c_i32, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_i64, c_i32, c_void_p]),
    "pcc_gather_rows": (c_int, [c_void_p, c_i32, c_void_p, c_i64, c_void_p, c_i32, c_void_p]),
    "pcc_scatter_rows": (c_int, [c_void_p, c_i32, c_void_p, c_i64, c_void_p, c_void_p]),
    "pcc_scatter_add_rows": (c_int, [c_void_p, c_i32, c_void_p, c_i64, c_void_p, c_void_p]),
    "pcc_compact_rows": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_topk_state_elems": (c_i64, [c_i32]),
    "pcc_topk_mask": (c_int, [c_void_p, c_i32, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_count_per_batch": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "pcc_sort_scratch_bytes": (c_i64, [c_i64]),
    "pcc_sort_coords": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_kernel_map_transpose": (c_int, [c_void_p, c_i64, c_i32, c_i64, c_void_p, c_void_p, c_void_p]),
    "pcc_conv_wgrad_scratch_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "pcc_conv_wgrad": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
                               c_i64, c_void_p]),
    "pcc_conv_wgrad_bf16": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
                                    c_i64, c_void_p]),
    "pcc_octree_scratch_bytes": (c_i64, [c_i64]),
    "pcc_octree_occupancy": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_octree_expand": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_i32, c_i64, c_void_p, c_void_p, c_i64, c_void_p]),
    "pcc_nn_search": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    "pcc_eb_quantize": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_eb_dequantize": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_eb_likelihood": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_gc_encode_prep": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_gc_dequantize": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "pcc_gc_encode_prep_packed": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_gc_dequantize_i16": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "pcc_rans_encode_with_indexes_i16u8": (c_i64, [c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i64]),
    "pcc_rans_decode_with_indexes_u8i16": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pcc_gc_forward": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_rans_encode_with_indexes": (c_i64, [c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i64]),
    "pcc_rans_decode_with_indexes": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_void_p]),
    "pcc_pmf_to_quantized_cdf": (c_int, [c_void_p, c_i32, c_i32, c_void_p]),
}


def build(force=False):
    """Compile libpcc_hip.so for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    check_kernel_resources()
    check_small_kernel_lds_reads()
    return SO_PATH


# kernels that must run out of registers alone: the MFMA convolutions (a spilled accumulator is a 20x slowdown, and
# conv_small_kernel's inline-assembly LDS reads sit behind a hand-placed s_waitcnt that the compiler does not see: a spilled
# or copied operand register would be read before its data has landed, silently)
NO_SCRATCH_KERNELS = ("conv_small_kernel", "conv_mfma_buf_kernel", "conv_mfma_kernel")


def check_small_kernel_lds_reads(obj=None):
    """conv_small_kernel reads its MFMA operands with inline-assembly ds_read_b128, which the compiler does not count in
    lgkmcnt; the kernel waits for them with a hand-placed s_waitcnt (csrc/conv.hip, pick()).  That is correct only while no
    instruction touches a destination register of such a read between the read and the wait — a register copy or spill the
    compiler inserted there would move stale data, silently.  This lints the generated code: the device code object
    is extracted from build/conv.o, disassembled, and every conv_small_kernel instantiation is walked; any mention of a pending
    read's registers before the next `s_waitcnt lgkmcnt(0)` fails the build.  -> number of reads checked."""
    import glob
    import re
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = obj or os.path.join(_HERE, "build", "conv.o")
    if not os.path.exists(objdump) or not os.path.exists(obj):
        raise RuntimeError(f"cannot lint conv_small_kernel: {objdump} or {obj} is missing")
    tmp = os.path.join(_HERE, "build", "_lint")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    local = os.path.join(tmp, "conv.o")
    shutil.copy(obj, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)
    cos = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
    if not cos:
        raise RuntimeError("cannot lint conv_small_kernel: no device code object in build/conv.o")
    dis = subprocess.run([objdump, "-d", cos[0]], check=True, capture_output=True, text=True).stdout
    shutil.rmtree(tmp, ignore_errors=True)
    return lint_lds_reads(dis)


def lint_lds_reads(dis, kernel="conv_small_kernel"):
    """the walk of check_small_kernel_lds_reads over a disassembly listing (llvm-objdump -d): -> reads checked, or RuntimeError"""
    import re

    def regs(operand):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", operand)
        return {int(m.group(1))} if m else set()

    checked, func, pending = 0, None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func = m.group(1) if kernel in m.group(1) else None
            pending = {}
            continue
        if func is None:
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        parts = text.split(None, 1)
        mnem = parts[0]
        ops = [o.strip().split(" ")[0] for o in parts[1].split(",")] if len(parts) > 1 else []
        if mnem == "s_waitcnt" and "lgkmcnt(0)" in text:
            pending = {}
            continue
        touched = set()
        for o in ops:
            touched |= regs(o)
        clash = touched & set(pending)
        if mnem == "ds_read_b128":
            dest = regs(ops[0])
            clash = (touched - dest) & set(pending) | (dest & set(pending))
            if not clash:
                for r_ in dest:
                    pending[r_] = text
                checked += 1
        if clash:
            raise RuntimeError(f"{func}: `{text}` touches v{sorted(clash)} while an LDS read into them is still in flight "
                               f"(`{pending[sorted(clash)[0]]}`): the hand-placed s_waitcnt of conv_small_kernel no longer covers it")
    if not checked:
        raise RuntimeError(f"cannot lint {kernel}: no ds_read_b128 found in its disassembly")
    return checked


def check_kernel_resources(path=None):
    """Parse the compiler's per-kernel resource remarks (csrc/Makefile writes them beside the objects) and fail loudly
    when one of NO_SCRATCH_KERNELS uses scratch memory or spills registers.  -> {mangled kernel name: (vgprs, scratch)}"""
    import re
    path = path or os.path.join(_HERE, "build", "conv.resources.txt")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the library was not built by csrc/Makefile")
    seen, name = {}, None
    vals = {}
    with open(path) as f:
        for line in f:
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                name, vals = m.group(1), {}
                seen[name] = vals
                continue
            m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
            if m and name:
                vals[m.group(1)] = int(m.group(2))
    bad = {k: v for k, v in seen.items() if any(t in k for t in NO_SCRATCH_KERNELS)
           and (v.get("ScratchSize [bytes/lane]", 0) or v.get("SGPRs Spill", 0) or v.get("VGPRs Spill", 0))}
    checked = [k for k in seen if any(t in k for t in NO_SCRATCH_KERNELS)]
    if not checked:
        raise RuntimeError(f"{path}: no resource remarks for {NO_SCRATCH_KERNELS} (compiler flag dropped?)")
    if bad:
        raise RuntimeError(f"kernels with scratch memory or spills (must be none): {bad}")
    return {k: (seen[k].get("VGPRs"), seen[k].get("ScratchSize [bytes/lane]")) for k in checked}


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the codec operators.")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class PccError(RuntimeError):
    pass


def check(rc):
    if rc < 0:
        raise PccError(f"libpcc_hip: error {rc}: {lib().pcc_last_error().decode(errors='replace')}")
    return rc


def ptr(t):
    """Device (or host) pointer of a tensor / numpy array, or NULL."""
    if t is None:
        return None
    try:
        return t.data_ptr()
    except AttributeError:
        return t.ctypes.data


_raw_stream = None


def stream():
    """Raw handle of the calling thread's current HIP stream on the current device.  Asked for at every launch (~230 times
    per frame): torch.cuda.current_stream() builds a Stream object through several Python layers (8.6 us, 2 ms per
    frame); the two C calls below return the same handle in well under a microsecond."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        try:
            get_stream, get_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
            get_stream(get_dev())
            _raw_stream = lambda: get_stream(get_dev())
        except AttributeError:                      # a torch build without the private accessors
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _raw_stream()
