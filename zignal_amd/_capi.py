"""ctypes binding of libflowz_hip.so (C ABI: include/flowz_hip.h).

The HIP library is THE product path: if it is missing or fails to load, importing this module
raises -- there is no CPU or PyTorch fallback anywhere in zignal_amd.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libflowz_hip.so")


class FlowzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[fz_status {code}] {msg}")
        self.code = code


class NoDeviceError(FlowzError):
    pass


FZ_OK, FZ_E_INVALID, FZ_E_GRAPH, FZ_E_NO_DEVICE, FZ_E_HIP, FZ_E_COMPILE, FZ_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
FZ_OP_ADD, FZ_OP_SUB, FZ_OP_MUL, FZ_OP_DIV, FZ_OP_NEG = 1, 2, 3, 4, 5
FZ_VF_STAGE_PACK, FZ_VF_NO_STAGE_PACK, FZ_VF_OUT_F64 = 8, 16, 64
FZ_OP_LT, FZ_OP_LE, FZ_OP_GT, FZ_OP_GE, FZ_OP_EQ, FZ_OP_NE, FZ_OP_NOT, FZ_OP_AND, FZ_OP_OR = 6, 7, 8, 9, 10, 11, 12, 13, 14
FZ_VF_PREFETCH3 = 32
FZ_VF_STREAM_MAJOR = 128
FZ_VF_SM_LONG, FZ_VF_SM_SHORT, FZ_VF_WAVE_SPLIT, FZ_VF_IO_WAVE = 256, 512, 1024, 32768
FZ_VF_LOCKSTEP = 524288
FZ_VF_GRID_SYNC = 8388608
FZ_VF_IO_WAVE2 = 33554432


def FZ_VF_WAVES(n):
    """wave split into n = 2, 3 or 4 parts"""
    return (n - 1) << 10 if 2 <= n <= 4 else 0


def FZ_VF_MAX_WG(n):
    """at most n workgroups per CU (flags bits 20..22)"""
    return (int(n) & 7) << 20
IR_KINDS = {1: "input", 2: "const", 3: "param", 4: "delay", 5: "add", 6: "sub", 7: "mul", 8: "div", 9: "neg", 10: "widen", 11: "narrow", 12: "mod", 13: "abslt", 14: "select", 15: "lt", 16: "le", 17: "gt", 18: "ge", 19: "eq", 20: "ne"}
FZ_DT_F32, FZ_DT_F64, FZ_DT_CF32, FZ_DT_CF64 = 0, 1, 2, 3
DTYPES = {"f32": 0, "f64": 1, "cf32": 2, "cf64": 3}


class Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ("n_in", "n_out", "n_nodes", "n_ops", "n_lines", "n_state", "n_const", "n_param", "max_delay", "n_lds_slots",
                 "stage_packable", "n_const64", "n_out_wires", "n_in_wires", "typed", "n_mod", "differs_from_reference")]


class IrNode(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("a", ctypes.c_uint32), ("b", ctypes.c_uint32), ("value", ctypes.c_float),
                ("dtype", ctypes.c_uint32), ("value64", ctypes.c_double), ("c", ctypes.c_uint32)]


class KernelResources(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("vgprs", "agprs", "sgprs", "scratch_bytes", "lds_bytes", "vgpr_spills", "sgpr_spills", "unroll")]


class Variant(ctypes.Structure):
    _fields_ = [("streams_per_lane", ctypes.c_uint32), ("unroll", ctypes.c_uint32),
                ("block_threads", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C zignal_amd/csrc`). zignal_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    P, u32, u64, f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float
    sig = {
        "fz_last_error": (ctypes.c_char_p, []),
        "fz_version": (ctypes.c_char_p, []),
        "fz_placeholder": (P, [u32]),
        "fz_delayed": (P, [u32, u32]),
        "fz_literal": (P, [f32]),
        "fz_literal_f64": (P, [ctypes.c_double]),
        "fz_literal_c32": (P, [ctypes.c_float, ctypes.c_float]),
        "fz_literal_c64": (P, [ctypes.c_double, ctypes.c_double]),
        "fz_stream_param": (P, [u32]),
        "fz_modulator": (P, [u32]),
        "fz_program_set_modulation": (ctypes.c_int, [P, P, u32]),
        "fz_uniform": (P, [u32, f32]),
        "fz_program_set_uniform": (ctypes.c_int, [P, u32, f32]),
        "fz_arith": (P, [ctypes.c_int, P, P]),
        "fz_channel": (P, [P, P]),
        "fz_parallel": (P, [P, P]),
        "fz_sequence": (P, [P, P]),
        "fz_feedback": (P, [P]),
        "fz_expr_retain": (None, [P]),
        "fz_expr_release": (None, [P]),
        "fz_input_arity": (ctypes.c_int, [P]),
        "fz_output_arity": (ctypes.c_int, [P]),
        "fz_max_input_delays": (ctypes.c_int, [P, ctypes.POINTER(u32), u32]),
        "fz_compile": (ctypes.c_int, [P, ctypes.POINTER(P)]),
        "fz_compile_typed": (ctypes.c_int, [P, ctypes.POINTER(u32), u32, ctypes.POINTER(P)]),
        "fz_program_input_dtypes": (ctypes.c_int, [P, ctypes.POINTER(u32), u32]),
        "fz_program_line_dtypes": (ctypes.c_int, [P, ctypes.POINTER(u32), u32]),
        "fz_program_destroy": (None, [P]),
        "fz_program_info": (ctypes.c_int, [P, ctypes.POINTER(Info)]),
        "fz_program_ir": (ctypes.c_int, [P, ctypes.POINTER(IrNode), u32]),
        "fz_program_outputs": (ctypes.c_int, [P, ctypes.POINTER(u32), u32]),
        "fz_program_output_dtypes": (ctypes.c_int, [P, ctypes.POINTER(u32), u32]),
        "fz_program_lines": (ctypes.c_int, [P, ctypes.POINTER(u32), ctypes.POINTER(u32), u32]),
        "fz_program_get_const": (ctypes.c_int, [P, u32, ctypes.POINTER(f32)]),
        "fz_program_set_const": (ctypes.c_int, [P, u32, f32]),
        "fz_program_build": (ctypes.c_int, [P, ctypes.POINTER(Variant)]),
        "fz_program_build_for": (ctypes.c_int, [P, ctypes.POINTER(Variant), u64, u32, u32]),
        "fz_program_wave_part": (ctypes.c_int, [P, u32, u32, ctypes.POINTER(P)]),
        "fz_program_kernel_resources": (ctypes.c_int, [P, ctypes.POINTER(Variant), u64, u32, u32, ctypes.c_int, ctypes.POINTER(KernelResources)]),
        "fz_program_kernel_name": (ctypes.c_long, [P, ctypes.POINTER(Variant), u64, u32, u32, ctypes.c_char_p, ctypes.c_size_t]),
        "fz_expr_recipe": (ctypes.c_long, [P, ctypes.c_char_p, ctypes.c_size_t]),
        "fz_expr_from_recipe": (P, [ctypes.c_char_p]),
        "fz_manifest_build": (ctypes.c_int, [ctypes.c_char_p, u32, ctypes.POINTER(u32)]),
        "fz_program_kernel_code_id": (ctypes.c_long, [P, ctypes.POINTER(Variant), u64, u32, u32, ctypes.c_char_p, ctypes.c_size_t]),
        "fz_program_kernel_symbol": (ctypes.c_long, [P, ctypes.POINTER(Variant), u64, u32, u32, ctypes.c_char_p, ctypes.c_size_t]),
        "fz_program_source": (ctypes.c_long, [P, ctypes.POINTER(Variant), ctypes.c_char_p, ctypes.c_size_t]),
        "fz_run_block": (ctypes.c_int, [P, P, P, P, P, u64, u32, ctypes.POINTER(Variant), P]),
        "fz_run_block_tiled": (ctypes.c_int, [P, P, P, P, P, u64, u32, u32, ctypes.POINTER(Variant), P]),
        "fz_run_block_window": (ctypes.c_int, [P, P, P, P, P, u64, u32, u32, u32, u32, ctypes.POINTER(Variant), P]),
        "fz_run_block_stream_major": (ctypes.c_int, [P, P, P, P, P, u64, u32, u32, u32, ctypes.POINTER(Variant), P]),
        "fz_bank_process_stream_major": (ctypes.c_int, [P, P, P, u32, u32, u32, ctypes.POINTER(Variant), P]),
        "fz_bank_process_blocks": (ctypes.c_int, [P, P, P, u32, u32, P, u32, ctypes.POINTER(Variant), P]),
        "fz_program_tune": (ctypes.c_int, [P, P, P, P, P, u64, u32, u32, P, ctypes.POINTER(Variant), ctypes.POINTER(f32)]),
        "fz_program_plan": (ctypes.c_int, [P, u64, u32, ctypes.POINTER(Variant)]),
        "fz_program_tune_candidates": (ctypes.c_int, [P, u64, u32, u32, ctypes.POINTER(Variant), u32]),
        "fz_recommended_tile_streams": (u32, [P]),
        "fz_bank_create": (ctypes.c_int, [P, u64, ctypes.POINTER(P)]),
        "fz_bank_clone": (ctypes.c_int, [P, ctypes.POINTER(P)]),
        "fz_bank_destroy": (None, [P]),
        "fz_bank_reset": (ctypes.c_int, [P]),
        "fz_bank_set_params_host": (ctypes.c_int, [P, P]),
        "fz_bank_state_device": (P, [P]),
        "fz_bank_process": (ctypes.c_int, [P, P, P, u32, ctypes.POINTER(Variant), P]),
        "fz_bank_process_tiled": (ctypes.c_int, [P, P, P, u32, u32, ctypes.POINTER(Variant), P]),
        "fz_bank_tune": (ctypes.c_int, [P, P, P, u32, u32, P, ctypes.POINTER(Variant), ctypes.POINTER(f32)]),
        "fz_bank_process_host": (ctypes.c_int, [P, P, P, u32]),
        "fz_bank_process_host_stream_major": (ctypes.c_int, [P, P, P, u32]),
        "fz_bank_process_host_f64": (ctypes.c_int, [P, P, P, u32]),
        "fz_device_count": (ctypes.c_int, []),
        "fz_synth_fill": (ctypes.c_int, [P, u64, u32, u32, u32, u64, u64, u32, P]),
        "fz_rbj_lowpass": (ctypes.c_int, [P, P, f32, u64, P, P, P]),
        "fz_copy_probe": (ctypes.c_int, [P, P, u64, P]),
        "fz_transpose_frames": (ctypes.c_int, [P, P, u64, u32, u32, u32, ctypes.c_int, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)       # AttributeError here == the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTS = _load()


def last_error() -> str:
    return (lib.fz_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> int:
    if rc < 0:
        cls = NoDeviceError if rc == FZ_E_NO_DEVICE else FlowzError
        raise cls(rc, last_error())
    return rc


def check_ptr(p):
    if not p:
        raise FlowzError(FZ_E_INVALID, last_error())
    return p
