"""ctypes binding of include/ahsoka_phase.h (the C-ABI of the HIP phasing path).

This is plumbing: it mirrors the C structs one to one and copies results into
numpy arrays.  There is no Python implementation of any stage here and no
fallback: if ``libahsoka_phase.so`` has not been built (``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C ahsoka_amd/csrc``) importing
the library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libahsoka_phase.so")

# status codes (include/ahsoka_phase.h)
ST_PHASED, ST_SKIPPED_LE1, ST_EMPTY_READSET, ST_NO_FULL_MATCH, ST_UNSUPPORTED = range(5)
E_ARG, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED, E_IO, E_PARSE = -1, -2, -3, -4, -5, -6, -7

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)


class CChainBatch(C.Structure):
    _fields_ = [
        ("n_chains", C.c_int32),
        ("chain_id", i32p),
        ("bubble_off", i64p),
        ("allele_off", i64p),
        ("apath_off", i64p),
        ("apath_node", i32p),
        ("apath_sign", u8p),
        ("genotype", i32p),
        ("node_off", i64p),
        ("node_id", i32p),
        ("read_off", i64p),
        ("rnode_off", i64p),
        ("rnode", i32p),
        ("read_mapq", i32p),
        ("read_name", i32p),
        ("read_flags", u8p),
        ("s1_order", i32p),
    ]


class CPhaseParams(C.Structure):
    _fields_ = [
        ("ploidy", C.c_int32),
        ("min_mapq", C.c_int32),
        ("min_variants", C.c_int32),
        ("min_overlap", C.c_int32),
        ("switch_cost", C.c_double),
        ("affine_switch_cost", C.c_double),
        ("compaction_mode", C.c_int32),
        ("device", C.c_int32),
        ("want_stages", C.c_int32),
        ("max_slots", C.c_int32),
        ("symmetry_opt", C.c_int32),
        ("row_limit", C.c_int32),
        ("wave_mode", C.c_int32),
    ]


class CPhaseResult(C.Structure):
    _fields_ = [
        ("n_chains", C.c_int32),
        ("ploidy", C.c_int32),
        ("status", i32p),
        ("pos_off", i64p),
        ("pos", i32p),
        ("hap_allele", i32p),
        ("hap_cluster", i32p),
        ("n_reads", i32p),
        ("n_clusters", i32p),
        ("n_iterations", i64p),
        ("vmat_off", i64p),
        ("vfull", i8p),
        ("vpart", i8p),
        ("fread_off", i64p),
        ("fread_name", i32p),
        ("fread_cluster", i32p),
        ("score_off", i64p),
        ("score_q", i32p),
        ("elapsed_ms", C.c_double),
        ("kernel_ms", C.c_double),
        ("_owner", C.c_void_p),
    ]


class CSynthSpec(C.Structure):
    _fields_ = [
        ("n_chains", C.c_int32),
        ("n_bubbles", C.c_int32),
        ("n_reads", C.c_int32),
        ("ploidy", C.c_int32),
        ("b_mode", C.c_int32),
        ("mean_span", C.c_double),
        ("allele_err", C.c_double),
        ("seed", C.c_uint64),
        ("first_chain", C.c_int32),
    ]


# every symbol include/ahsoka_phase.h declares (tests check the library exports them all)
ABI_SYMBOLS = [
    "ahsoka_default_params", "ahsoka_backend_info", "ahsoka_abi_version", "ahsoka_phase_batch", "ahsoka_phase_batch_multi",
    "ahsoka_result_free", "ahsoka_batch_upload", "ahsoka_phase_device", "ahsoka_batch_download",
    "ahsoka_batch_free", "ahsoka_batch_result_device", "ahsoka_batch_kernel_times", "ahsoka_batch_bytes", "ahsoka_render_chain",
    "ahsoka_ingest", "ahsoka_ingest_dump", "ahsoka_batch_host_free", "ahsoka_synth_batch", "ahsoka_synth_select", "ahsoka_synth_shapes", "ahsoka_synth_write_text", "ahsoka_synth_truth",
    "ahsoka_test_introsort_u32", "ahsoka_test_introsort_par_u32", "ahsoka_test_detlog",
    "ahsoka_test_multi_plan", "ahsoka_kernel_revision", "ahsoka_batch_class_info",
]

_lib = None


def lib() -> C.CDLL:
    """The HIP library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (make -C ahsoka_amd/csrc, or "
            "__graft_entry__.build()). ahsoka_amd has no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    L.ahsoka_default_params.argtypes = [C.POINTER(CPhaseParams)]
    L.ahsoka_default_params.restype = None
    L.ahsoka_backend_info.argtypes = [C.c_char_p, C.c_size_t]
    L.ahsoka_kernel_revision.argtypes = []
    L.ahsoka_kernel_revision.restype = C.c_char_p
    L.ahsoka_phase_batch.argtypes = [C.POINTER(CChainBatch), C.POINTER(CPhaseParams), C.POINTER(CPhaseResult)]
    L.ahsoka_result_free.argtypes = [C.POINTER(CPhaseResult)]
    L.ahsoka_result_free.restype = None
    L.ahsoka_phase_batch_multi.argtypes = [C.POINTER(CChainBatch), C.POINTER(CPhaseParams), i32p, C.c_int32,
                                           C.POINTER(CPhaseResult)]
    L.ahsoka_batch_upload.argtypes = [C.POINTER(CChainBatch), C.POINTER(CPhaseParams), C.POINTER(C.c_void_p)]
    L.ahsoka_phase_device.argtypes = [C.c_void_p]
    L.ahsoka_batch_download.argtypes = [C.c_void_p, C.POINTER(CPhaseResult)]
    L.ahsoka_batch_free.argtypes = [C.c_void_p]
    L.ahsoka_batch_free.restype = None
    L.ahsoka_batch_result_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.ahsoka_batch_class_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
    L.ahsoka_batch_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_char_p)),
                                            C.POINTER(C.POINTER(C.c_float))]
    L.ahsoka_batch_bytes.argtypes = [C.POINTER(CChainBatch), C.c_int32, C.POINTER(C.c_int64)]
    L.ahsoka_render_chain.argtypes = [C.POINTER(CChainBatch), C.POINTER(CPhaseResult), C.c_int32, C.c_char_p, C.c_int64]
    L.ahsoka_render_chain.restype = C.c_int64
    L.ahsoka_ingest.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(CChainBatch))]
    L.ahsoka_ingest_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.ahsoka_batch_host_free.argtypes = [C.POINTER(CChainBatch)]
    L.ahsoka_batch_host_free.restype = None
    L.ahsoka_synth_batch.argtypes = [C.POINTER(CSynthSpec), C.POINTER(C.POINTER(CChainBatch))]
    L.ahsoka_synth_select.argtypes = [C.POINTER(CSynthSpec), i32p, C.c_int32, C.POINTER(C.POINTER(CChainBatch))]
    L.ahsoka_synth_shapes.argtypes = [C.POINTER(CSynthSpec), i32p, i32p]
    L.ahsoka_synth_write_text.argtypes = [C.POINTER(CSynthSpec), C.c_char_p, C.c_char_p]
    L.ahsoka_synth_truth.argtypes = [C.POINTER(CSynthSpec), u8p, C.c_int64]
    L.ahsoka_synth_truth.restype = C.c_int64
    L.ahsoka_test_introsort_u32.argtypes = [C.POINTER(C.c_uint32), C.c_int32]
    L.ahsoka_test_introsort_par_u32.argtypes = [C.POINTER(C.c_uint32), C.c_int32]
    L.ahsoka_test_detlog.argtypes = [C.c_double]
    L.ahsoka_test_detlog.restype = C.c_double
    L.ahsoka_test_multi_plan.argtypes = [C.POINTER(CChainBatch), C.POINTER(CPhaseParams), C.c_int32, i32p, i32p,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32]
    _lib = L
    return L


class AhsokaError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what} failed with code {code}")
        self.code = code


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise AhsokaError(rc, what)


def default_params(**over) -> CPhaseParams:
    p = CPhaseParams()
    lib().ahsoka_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def backend_info() -> str:
    buf = C.create_string_buffer(128)
    lib().ahsoka_backend_info(buf, 128)
    return buf.value.decode()


def kernel_revision() -> str:
    return lib().ahsoka_kernel_revision().decode()


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class ChainBatch:
    """A ChainBatch in host memory: either owned by the library (synthetic
    generator, ingest) or assembled from numpy arrays."""

    def __init__(self, cptr=None, arrays: Optional[Dict[str, np.ndarray]] = None):
        self._cptr = cptr
        self._keep = None
        if arrays is not None:
            self._keep = {}
            cb = CChainBatch()
            cb.n_chains = int(len(arrays["chain_id"]))
            for name, ctype, npdt in _FIELDS:
                a = arrays.get(name)
                if a is None:
                    setattr(cb, name, C.cast(None, C.POINTER(ctype)))
                    continue
                a = np.ascontiguousarray(a, dtype=npdt)
                if a.size == 0:
                    a = np.zeros(1, dtype=npdt)
                self._keep[name] = a
                setattr(cb, name, a.ctypes.data_as(C.POINTER(ctype)))
            self._struct = cb

    @property
    def c(self):
        return self._cptr.contents if self._cptr is not None else self._struct

    def ptr(self):
        return self._cptr if self._cptr is not None else C.pointer(self._struct)

    @property
    def n_chains(self) -> int:
        return int(self.c.n_chains)

    def arrays(self) -> Dict[str, Optional[np.ndarray]]:
        c = self.c
        Cn = c.n_chains
        out: Dict[str, Optional[np.ndarray]] = {}
        out["chain_id"] = _np(c.chain_id, Cn, np.int32)
        out["bubble_off"] = _np(c.bubble_off, Cn + 1, np.int64)
        SB = int(out["bubble_off"][-1]) if Cn else 0
        out["allele_off"] = _np(c.allele_off, SB + 1, np.int64)
        SA = int(out["allele_off"][-1]) if Cn else 0
        out["apath_off"] = _np(c.apath_off, SA + 1, np.int64)
        SP = int(out["apath_off"][-1]) if Cn else 0
        out["apath_node"] = _np(c.apath_node, SP, np.int32)
        out["apath_sign"] = _np(c.apath_sign, SP, np.uint8) if c.apath_sign else None
        out["genotype"] = _np(c.genotype, SA, np.int32) if c.genotype else None
        out["node_off"] = _np(c.node_off, Cn + 1, np.int64)
        out["node_id"] = _np(c.node_id, int(out["node_off"][-1]) if Cn else 0, np.int32)
        out["read_off"] = _np(c.read_off, Cn + 1, np.int64)
        SR = int(out["read_off"][-1]) if Cn else 0
        out["rnode_off"] = _np(c.rnode_off, SR + 1, np.int64)
        out["rnode"] = _np(c.rnode, int(out["rnode_off"][-1]) if Cn else 0, np.int32)
        out["read_mapq"] = _np(c.read_mapq, SR, np.int32)
        out["read_name"] = _np(c.read_name, SR, np.int32)
        out["read_flags"] = _np(c.read_flags, SR, np.uint8) if c.read_flags else None
        out["s1_order"] = _np(c.s1_order, SB, np.int32) if c.s1_order else None
        return out

    def algorithmic_input_bytes(self, ploidy: int = 2) -> int:
        v = C.c_int64(0)
        _check(lib().ahsoka_batch_bytes(self.ptr(), ploidy, C.byref(v)), "ahsoka_batch_bytes")
        return int(v.value)

    def free(self):
        if self._cptr is not None:
            lib().ahsoka_batch_host_free(self._cptr)
            self._cptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_FIELDS = [
    ("chain_id", C.c_int32, np.int32), ("bubble_off", C.c_int64, np.int64), ("allele_off", C.c_int64, np.int64),
    ("apath_off", C.c_int64, np.int64), ("apath_node", C.c_int32, np.int32), ("apath_sign", C.c_uint8, np.uint8),
    ("genotype", C.c_int32, np.int32), ("node_off", C.c_int64, np.int64), ("node_id", C.c_int32, np.int32),
    ("read_off", C.c_int64, np.int64), ("rnode_off", C.c_int64, np.int64), ("rnode", C.c_int32, np.int32),
    ("read_mapq", C.c_int32, np.int32), ("read_name", C.c_int32, np.int32), ("read_flags", C.c_uint8, np.uint8),
    ("s1_order", C.c_int32, np.int32),
]

# BASELINE.json configs -> generator specs; seed = 0xA450CA00 + config number
CONFIGS = {
    "c1": dict(n_chains=10, n_bubbles=5, n_reads=50, ploidy=2, b_mode=0, mean_span=7.0, seed=0xA450CA01),
    "c2": dict(n_chains=1000, n_bubbles=20, n_reads=500, ploidy=2, b_mode=0, mean_span=7.0, seed=0xA450CA02),
    "c3": dict(n_chains=10000, n_bubbles=50, n_reads=1000, ploidy=2, b_mode=0, mean_span=7.0, seed=0xA450CA03),
    "c4": dict(n_chains=2000, n_bubbles=30, n_reads=2000, ploidy=4, b_mode=0, mean_span=7.0, seed=0xA450CA04),
    "c5": dict(n_chains=50000, n_bubbles=0, n_reads=0, ploidy=2, b_mode=1, mean_span=10.0, seed=0xA450CA05),
}


def synth_spec(n_chains, n_bubbles, n_reads, ploidy=2, b_mode=0, mean_span=7.0, allele_err=0.02, seed=0xA450CA00,
               first_chain=0) -> CSynthSpec:
    s = CSynthSpec()
    s.n_chains, s.n_bubbles, s.n_reads, s.ploidy = n_chains, n_bubbles, n_reads, ploidy
    s.b_mode, s.mean_span, s.allele_err, s.seed, s.first_chain = b_mode, mean_span, allele_err, seed, first_chain
    return s


def synth_batch(**kw) -> ChainBatch:
    spec = synth_spec(**kw)
    out = C.POINTER(CChainBatch)()
    _check(lib().ahsoka_synth_batch(C.byref(spec), C.byref(out)), "ahsoka_synth_batch")
    return ChainBatch(cptr=out)


def synth_select(chains, **kw) -> ChainBatch:
    """The listed global chains of a synthetic workload (n_chains / first_chain of kw are ignored)."""
    spec = synth_spec(**kw)
    idx = np.ascontiguousarray(chains, dtype=np.int32)
    out = C.POINTER(CChainBatch)()
    _check(lib().ahsoka_synth_select(C.byref(spec), idx.ctypes.data_as(i32p), len(idx), C.byref(out)), "ahsoka_synth_select")
    return ChainBatch(cptr=out)


def synth_shapes(**kw):
    """(n_bubbles, n_reads) of chains [first_chain, first_chain + n_chains) without generating them."""
    spec = synth_spec(**kw)
    nb = np.zeros(max(1, spec.n_chains), dtype=np.int32)
    nr = np.zeros(max(1, spec.n_chains), dtype=np.int32)
    _check(lib().ahsoka_synth_shapes(C.byref(spec), nb.ctypes.data_as(i32p), nr.ctypes.data_as(i32p)), "ahsoka_synth_shapes")
    return nb[:spec.n_chains], nr[:spec.n_chains]


def synth_truth(**kw) -> np.ndarray:
    """Truth alleles of the synthetic workload, flat: per chain [ploidy][B]."""
    spec = synth_spec(**kw)
    need = lib().ahsoka_synth_truth(C.byref(spec), C.cast(None, u8p), 0)
    if need < 0:
        raise AhsokaError(int(need), "ahsoka_synth_truth")
    buf = np.zeros(max(1, need), dtype=np.uint8)
    lib().ahsoka_synth_truth(C.byref(spec), buf.ctypes.data_as(u8p), need)
    return buf[:need]


def synth_write_text(gfa: str, gaf: str, **kw) -> None:
    spec = synth_spec(**kw)
    _check(lib().ahsoka_synth_write_text(C.byref(spec), gfa.encode(), gaf.encode()), "ahsoka_synth_write_text")


def ingest(gfa: str, gaf: str) -> ChainBatch:
    out = C.POINTER(CChainBatch)()
    _check(lib().ahsoka_ingest(gfa.encode(), gaf.encode(), C.byref(out)), "ahsoka_ingest")
    return ChainBatch(cptr=out)


def ingest_dump(gfa: str, gaf: str, out_path: str) -> None:
    _check(lib().ahsoka_ingest_dump(gfa.encode(), gaf.encode(), out_path.encode()), "ahsoka_ingest_dump")


@dataclass
class PhaseResult:
    """Numpy copy of an ahsoka_phase_result (same field meaning)."""
    ploidy: int
    status: np.ndarray
    pos_off: np.ndarray
    pos: np.ndarray
    hap_allele: np.ndarray
    hap_cluster: np.ndarray
    n_reads: np.ndarray
    n_clusters: np.ndarray
    n_iterations: np.ndarray
    elapsed_ms: float
    kernel_ms: float
    stages: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def n_chains(self) -> int:
        return len(self.status)

    def chain_haps(self, c: int) -> np.ndarray:
        """[ploidy][P] consensus alleles of chain c."""
        p0, p1 = int(self.pos_off[c]), int(self.pos_off[c + 1])
        P = p1 - p0
        return self.hap_allele[self.ploidy * p0: self.ploidy * p1].reshape(self.ploidy, P)

    def chain_pos(self, c: int) -> np.ndarray:
        return self.pos[int(self.pos_off[c]): int(self.pos_off[c + 1])]

    def phased_bubbles(self, bubble_off: np.ndarray) -> int:
        """BASELINE.json metric numerator: sum of B over chains with status PHASED."""
        b = np.diff(bubble_off)
        return int(b[self.status == ST_PHASED].sum())


def result_from_c(r: CPhaseResult) -> PhaseResult:
    Cn, p = r.n_chains, r.ploidy
    pos_off = _np(r.pos_off, Cn + 1, np.int64)
    SP = int(pos_off[-1]) if Cn else 0
    res = PhaseResult(
        ploidy=p,
        status=_np(r.status, Cn, np.int32),
        pos_off=pos_off,
        pos=_np(r.pos, SP, np.int32),
        hap_allele=_np(r.hap_allele, SP * p, np.int32),
        hap_cluster=_np(r.hap_cluster, SP * p, np.int32),
        n_reads=_np(r.n_reads, Cn, np.int32),
        n_clusters=_np(r.n_clusters, Cn, np.int32),
        n_iterations=_np(r.n_iterations, Cn, np.int64),
        elapsed_ms=float(r.elapsed_ms),
        kernel_ms=float(r.kernel_ms),
    )
    if r.vmat_off:
        vo = _np(r.vmat_off, Cn + 1, np.int64)
        fo = _np(r.fread_off, Cn + 1, np.int64)
        so = _np(r.score_off, Cn + 1, np.int64)
        res.stages = dict(
            vmat_off=vo, vfull=_np(r.vfull, int(vo[-1]), np.int8), vpart=_np(r.vpart, int(vo[-1]), np.int8),
            fread_off=fo, fread_name=_np(r.fread_name, int(fo[-1]), np.int32),
            fread_cluster=_np(r.fread_cluster, int(fo[-1]), np.int32),
            score_off=so, score_q=_np(r.score_q, int(so[-1]), np.int32),
        )
    return res


def phase_batch(batch: ChainBatch, params: Optional[CPhaseParams] = None) -> PhaseResult:
    """One call of the hot path: host ChainBatch in, host result out (HIP)."""
    prm = params if params is not None else default_params()
    r = CPhaseResult()
    _check(lib().ahsoka_phase_batch(batch.ptr(), C.byref(prm), C.byref(r)), "ahsoka_phase_batch")
    try:
        return result_from_c(r)
    finally:
        lib().ahsoka_result_free(C.byref(r))


def phase_batch_multi(batch: ChainBatch, devices, params: Optional[CPhaseParams] = None) -> PhaseResult:
    """The hot path over several HIP devices of this node (ahsoka_phase_batch_multi): static
    cost-weighted partition of the chains, one library-owned handle per device, results merged
    into the batch's chain order."""
    prm = params if params is not None else default_params()
    dv = np.ascontiguousarray(devices, dtype=np.int32)
    r = CPhaseResult()
    _check(lib().ahsoka_phase_batch_multi(batch.ptr(), C.byref(prm), dv.ctypes.data_as(i32p), len(dv), C.byref(r)),
           "ahsoka_phase_batch_multi")
    try:
        return result_from_c(r)
    finally:
        lib().ahsoka_result_free(C.byref(r))


def multi_plan(batch: ChainBatch, n_devices: int, params: Optional[CPhaseParams] = None, check: bool = True):
    """The static partition ahsoka_phase_batch_multi would use, without touching a device:
    (owner [C], blocks?, serial prologue ms, longest per-shard host ms)."""
    prm = params if params is not None else default_params()
    owner = np.zeros(max(1, batch.n_chains), dtype=np.int32)
    blocks = C.c_int32(0)
    serial, shard = C.c_double(0), C.c_double(0)
    _check(lib().ahsoka_test_multi_plan(batch.ptr(), C.byref(prm), n_devices, owner.ctypes.data_as(i32p), C.byref(blocks),
                                        C.byref(serial), C.byref(shard), 1 if check else 0), "ahsoka_test_multi_plan")
    return owner[:batch.n_chains], bool(blocks.value), float(serial.value), float(shard.value)


class DeviceBatch:
    """Batch resident in HBM: upload once, phase repeatedly (bench), download."""

    def __init__(self, batch: ChainBatch, params: Optional[CPhaseParams] = None):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        _check(lib().ahsoka_batch_upload(batch.ptr(), C.byref(self.params), C.byref(self._h)), "ahsoka_batch_upload")

    def phase(self) -> None:
        _check(lib().ahsoka_phase_device(self._h), "ahsoka_phase_device")

    def kernel_times_ms(self) -> Dict[str, float]:
        n = C.c_int32()
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        _check(lib().ahsoka_batch_kernel_times(self._h, C.byref(n), C.byref(names), C.byref(ms)), "kernel_times")
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def class_info(self) -> List[Dict[str, int]]:
        """size classes of the last phase() call: nmax, chains, slots, waves, workgroups per CU, LDS bytes, extras"""
        n = C.c_int32()
        buf = (C.c_int32 * (8 * 16))()
        _check(lib().ahsoka_batch_class_info(self._h, C.byref(n), buf, 8 * 16), "ahsoka_batch_class_info")
        keys = ("nmax", "chains", "slots", "waves", "workgroups_per_cu", "lds_bytes", "items_lds", "stage_row")
        return [dict(zip(keys, buf[8 * k:8 * k + 8])) for k in range(min(n.value, 16))]

    def result_device(self):
        """(device pointer, number of int32) of the packed result record block
        [status | npos | pos | hap_allele]; owned by this handle."""
        ptr = C.c_void_p()
        n = C.c_int64()
        _check(lib().ahsoka_batch_result_device(self._h, C.byref(ptr), C.byref(n)), "ahsoka_batch_result_device")
        return int(ptr.value), int(n.value)

    def result_tensor(self):
        """The packed result block as a torch int32 CUDA tensor (a view of library
        memory: valid until the next phase() / free())."""
        import torch

        ptr, n = self.result_device()

        class _Ext:  # __cuda_array_interface__ carrier
            pass

        e = _Ext()
        e.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        return torch.as_tensor(e, device=torch.device("cuda", self.params.device))

    def download(self) -> PhaseResult:
        r = CPhaseResult()
        _check(lib().ahsoka_batch_download(self._h, C.byref(r)), "ahsoka_batch_download")
        try:
            return result_from_c(r)
        finally:
            lib().ahsoka_result_free(C.byref(r))

    def free(self) -> None:
        if self._h:
            lib().ahsoka_batch_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def render_chain(batch: ChainBatch, result: PhaseResult, c: int) -> str:
    """Text of chain c as the reference writes it to <o>-result.txt."""
    # rebuild a C result view over the numpy arrays
    r = CPhaseResult()
    r.n_chains, r.ploidy = result.n_chains, result.ploidy
    keep = []

    def setp(name, arr, ct):
        a = np.ascontiguousarray(arr)
        if a.size == 0:
            a = np.zeros(1, dtype=a.dtype)
        keep.append(a)
        setattr(r, name, a.ctypes.data_as(C.POINTER(ct)))

    setp("status", result.status.astype(np.int32), C.c_int32)
    setp("pos_off", result.pos_off.astype(np.int64), C.c_int64)
    setp("pos", result.pos.astype(np.int32), C.c_int32)
    setp("hap_allele", result.hap_allele.astype(np.int32), C.c_int32)
    n = lib().ahsoka_render_chain(batch.ptr(), C.byref(r), c, None, 0)
    if n < 0:
        raise AhsokaError(int(n), "ahsoka_render_chain")
    buf = C.create_string_buffer(int(n) + 1)
    lib().ahsoka_render_chain(batch.ptr(), C.byref(r), c, buf, int(n) + 1)
    return buf.value.decode()
