"""ctypes binding of include/pheniqs_amd.h.  Plumbing only: it moves pointers, never computes."""
from __future__ import annotations

import ctypes
import json
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

RESULT_DTYPE = np.dtype([("index", "<i4"), ("distance", "u1"), ("hq_distance", "u1"), ("status", "u1"),
                         ("qcfail", "u1"), ("confidence", "<f8")])
OPTION_DTYPE = np.dtype([("count", "<u8"), ("pf_count", "<u8"), ("accumulated_distance", "<u8"),
                         ("accumulated_pf_distance", "<u8"), ("low_conditional_confidence_count", "<u8"),
                         ("low_confidence_count", "<u8"), ("accumulated_confidence", "<f8"),
                         ("accumulated_pf_confidence", "<f8")])

MODE_LITERAL = 0
MODE_FAST = 1
MODE_AUTO = 2
DEVICE_NONE = -1

# every symbol include/pheniqs_amd.h declares
SYMBOLS = ("phq_create", "phq_destroy", "phq_last_error", "phq_get_layout", "phq_get_decoder_info", "phq_set_mode",
           "phq_stage_reads", "phq_classify_device", "phq_classify_host", "phq_host_alloc", "phq_host_free",
           "phq_stats_reset", "phq_stats_device_pointer", "phq_stats_export", "phq_stats_import", "phq_stats_merge",
           "phq_stats_allreduce", "phq_summarize", "phq_stats_options", "phq_report_json", "phq_format_double",
           "phq_profile_enable", "phq_profile_collect", "phq_profile_chain", "phq_plan_chain", "phq_copy_bandwidth", "phq_device_path_counters", "phq_pack_planes", "phq_classify_host_packed", "phq_transcode_fastq_to_sam", "phq_transcode_job", "phq_transcode_job_copies", "phq_transcode_plan",
           "phq_compile_job", "phq_load_job", "phq_free_string", "phq_adjust_prior", "phq_build_id", "phq_qc_export", "phq_qc_merge")


class PheniqsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("[%d] %s" % (code, message))
        self.code = code


class Layout(ctypes.Structure):
    _fields_ = [("n_decoder", ctypes.c_int32), ("n_result", ctypes.c_int32), ("plane_width", ctypes.c_int32),
                ("nucleotide_width", ctypes.c_int32), ("stats_words", ctypes.c_int64),
                ("input_segment_cardinality", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class DecoderInfo(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in (
        "kind", "algorithm", "index", "n_barcode", "nucleotide_cardinality", "segment_cardinality",
        "plane_stride", "plane_prefix", "result_slot", "multiplexing")] + [("stats_offset", ctypes.c_int64)]


def library_path() -> str:
    """The in-tree library; PHQ_LIBRARY names another build of it (A/B runs of kernel variants on one box,
    profiles/tools/variant.sh)."""
    return os.environ.get("PHQ_LIBRARY") or os.path.join(HERE, "libpheniqs_amd.so")


_library = None


def load_library():
    """Loads the in-tree library.  No fallback: a missing build is an error."""
    global _library
    if _library is not None:
        return _library
    # One HIP runtime per process: PyTorch ships its own copy of libamdhip64 and has to get it in first.  If this
    # library pulled the system copy in first and torch then initialised its own, the second one to touch the device
    # reported "no ROCm-capable device" (seen on the MI355X pool).  torch is plumbing here, never required.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise PheniqsError(2, "libpheniqs_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`): " + path)
    L = ctypes.CDLL(path)
    P, I32, I64, U64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
    L.phq_create.argtypes = [ctypes.c_char_p, I32, ctypes.POINTER(P)]
    L.phq_destroy.argtypes = [P]
    L.phq_destroy.restype = None
    L.phq_last_error.argtypes = [P]
    L.phq_last_error.restype = ctypes.c_char_p
    L.phq_get_layout.argtypes = [P, ctypes.POINTER(Layout)]
    L.phq_get_decoder_info.argtypes = [P, I32, ctypes.POINTER(DecoderInfo)]
    L.phq_set_mode.argtypes = [P, I32]
    L.phq_stage_reads.argtypes = [P, I64, I32, P, P, P, P, P, I64, P, P, P, P, P, P, P]
    L.phq_classify_device.argtypes = [P, I64, I64, I64, P, P, P, P, P, P]
    L.phq_classify_host.argtypes = [P, I64, I64, P, P, P, P, P]
    L.phq_host_alloc.argtypes = [U64, ctypes.POINTER(P)]
    L.phq_host_free.argtypes = [P]
    L.phq_host_free.restype = None
    L.phq_stats_reset.argtypes = [P, P]
    L.phq_stats_device_pointer.argtypes = [P, ctypes.POINTER(P)]
    L.phq_stats_export.argtypes = [P, P]
    L.phq_stats_import.argtypes = [P, P]
    L.phq_stats_merge.argtypes = [P, P]
    L.phq_stats_allreduce.argtypes = [P, P, P]
    L.phq_summarize.argtypes = [P, ctypes.c_int64, ctypes.c_int64, P, P, P]
    L.phq_stats_options.argtypes = [P, P, I32, P]
    L.phq_report_json.argtypes = [P, P, U64, U64, ctypes.POINTER(ctypes.c_char_p)]
    L.phq_format_double.argtypes = [ctypes.c_double, I32, ctypes.c_char_p, I32]
    L.phq_transcode_fastq_to_sam.argtypes = [P, I32, P, ctypes.c_char_p, I64, ctypes.POINTER(U64), ctypes.POINTER(U64)]
    L.phq_transcode_job.argtypes = [P, I64, ctypes.POINTER(U64), ctypes.POINTER(U64)]
    L.phq_transcode_job_copies.argtypes = [P, I32, P, I64, ctypes.POINTER(U64), ctypes.POINTER(U64)]
    L.phq_transcode_plan.argtypes = [P, ctypes.POINTER(ctypes.c_void_p)]
    L.phq_profile_enable.argtypes = [P, I32]
    L.phq_profile_collect.argtypes = [P, P, P]
    L.phq_profile_chain.argtypes = [P, ctypes.POINTER(ctypes.c_void_p)]
    L.phq_plan_chain.argtypes = [P, I64, I32, ctypes.POINTER(ctypes.c_void_p)]
    L.phq_copy_bandwidth.argtypes = [I32, U64, I32, ctypes.POINTER(ctypes.c_double)]
    L.phq_device_path_counters.argtypes = [ctypes.POINTER(U64)]
    L.phq_pack_planes.argtypes = [P, I64, I64, P, P, P, P, I64, ctypes.POINTER(I64)]
    L.phq_classify_host_packed.argtypes = [P, I64, I64, P, P, I64, P, P, P]
    L.phq_compile_job.argtypes = [ctypes.c_char_p, ctypes.POINTER(P)]
    L.phq_load_job.argtypes = [ctypes.c_char_p, ctypes.POINTER(P)]
    L.phq_free_string.argtypes = [P]
    L.phq_free_string.restype = None
    L.phq_adjust_prior.argtypes = [P, P, ctypes.c_char_p, ctypes.POINTER(P)]
    L.phq_qc_export.argtypes = [P, P, I64, ctypes.POINTER(I64)]
    L.phq_qc_merge.argtypes = [P, P, I64]
    L.phq_build_id.argtypes = []
    L.phq_build_id.restype = ctypes.c_char_p
    _library = L
    return L


def build_id() -> str:
    """phq_build_id: SHA-256 over the sources the loaded library was built from."""
    return load_library().phq_build_id().decode()


def source_id() -> str:
    """The same hash computed from the tree (pheniqs_amd/csrc/Makefile, `build_id.inc`): equal to build_id() when the
    library is the build of the sources beside it."""
    import glob
    import hashlib
    csrc = os.path.join(HERE, "csrc")
    names = sorted([os.path.basename(p) for pattern in ("*.cpp", "*.hip", "*.hpp") for p in glob.glob(os.path.join(csrc, pattern))] + ["Makefile"])
    # make's $(sort) orders byte wise over the whole word list, the header's relative path included
    words = sorted(names + ["../../include/pheniqs_amd.h"])
    digest = hashlib.sha256()
    for word in words:
        digest.update(open(os.path.join(csrc, word), "rb").read())
    return digest.hexdigest()


def copy_bandwidth(device: int = 0, gigabytes: float = 1.0, repeats: int = 10) -> float:
    """phq_copy_bandwidth: GB/s (read + written) of a plain 16 bytes a lane streaming copy on the device."""
    out = ctypes.c_double(0)
    status = load_library().phq_copy_bandwidth(device, int(gigabytes * (1 << 30)) // 16 * 16, repeats, ctypes.byref(out))
    if status != 0:
        raise PheniqsError(status, "phq_copy_bandwidth")
    return out.value


def device_path_counters() -> Dict[str, int]:
    """phq_device_path_counters: what the transcoding calls of this process did on the device so far."""
    out = (ctypes.c_uint64 * 4)()
    status = load_library().phq_device_path_counters(out)
    if status != 0:
        raise PheniqsError(status, "phq_device_path_counters")
    return {"rounds": int(out[0]), "reads": int(out[1]), "record_bytes": int(out[2]), "resumed": int(out[3])}


def compile_job(job: Dict[str, Any]) -> Dict[str, Any]:
    """phq_compile_job: the library's configuration compiler (decoder topics) on a user level job document."""
    L = load_library()
    out = ctypes.c_void_p()
    status = L.phq_compile_job(json.dumps(job).encode(), ctypes.byref(out))
    if status != 0:
        raise PheniqsError(status, (L.phq_last_error(None) or b"").decode())
    try:
        return json.loads(ctypes.string_at(out.value).decode())
    finally:
        L.phq_free_string(out)


def load_job(path: str) -> Dict[str, Any]:
    """phq_load_job: a job document file with its `import` list resolved (job.cpp:163-224)."""
    L = load_library()
    out = ctypes.c_void_p()
    status = L.phq_load_job(os.fspath(path).encode(), ctypes.byref(out))
    if status != 0:
        raise PheniqsError(status, (L.phq_last_error(None) or b"").decode())
    try:
        return json.loads(ctypes.string_at(out.value).decode())
    finally:
        L.phq_free_string(out)


def format_double(value: float, precision: int = 15) -> str:
    buffer = ctypes.create_string_buffer(64)
    status = load_library().phq_format_double(value, precision, buffer, 64)
    if status != 0:
        raise PheniqsError(status, "phq_format_double")
    return buffer.value.decode()


def _pointer(x) -> Optional[int]:
    """numpy array -> host address, torch tensor -> data_ptr, int -> as is."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return int(x.data_ptr())


class _CudaArray:
    """Wraps a raw device pointer so torch.as_tensor can adopt it (__cuda_array_interface__)."""

    def __init__(self, pointer: int, words: int, owner):
        self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (pointer, False), "version": 2}
        self._owner = owner


class DecoderSet:
    """One TranscodingDecoder worth of decoders (reference transcode.h:40-77) on one device."""

    def __init__(self, job: Any, device: int = 0):
        self.lib = load_library()
        text = job if isinstance(job, str) else json.dumps(job)
        handle = ctypes.c_void_p()
        status = self.lib.phq_create(text.encode(), device, ctypes.byref(handle))
        if status != 0:
            raise PheniqsError(status, (self.lib.phq_last_error(None) or b"").decode())
        self.handle = handle
        self.device = device
        layout = Layout()
        self._check(self.lib.phq_get_layout(self.handle, ctypes.byref(layout)))
        self.n_decoder = layout.n_decoder
        self.n_result = layout.n_result
        self.plane_width = layout.plane_width
        self.nucleotide_width = layout.nucleotide_width
        self.stats_words = layout.stats_words
        self.decoders: List[Dict[str, int]] = []
        for k in range(self.n_decoder):
            info = DecoderInfo()
            self._check(self.lib.phq_get_decoder_info(self.handle, k, ctypes.byref(info)))
            self.decoders.append({name: getattr(info, name) for name, _ in DecoderInfo._fields_})

    def close(self):
        if getattr(self, "handle", None):
            self.lib.phq_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status: int):
        if status != 0:
            raise PheniqsError(status, (self.lib.phq_last_error(self.handle) or b"").decode())

    def set_mode(self, mode: int):
        self._check(self.lib.phq_set_mode(self.handle, mode))

    # ---- layout helpers -----------------------------------------------------------------------------

    def planes_from_rows(self, rows: np.ndarray, offsets: Sequence[int], capacity: Optional[int] = None) -> np.ndarray:
        """Row major observations [n, sum L] (the oracle's staging) -> decoder major planes (flat uint8) laid out for
        `capacity` reads (default n)."""
        n = rows.shape[0]
        capacity = n if capacity is None else capacity
        out = np.zeros(capacity * max(self.plane_width, 1), dtype=np.uint8)
        for d, offset in zip(self.decoders, offsets):
            if d["plane_stride"] == 0:
                continue
            plane = out[capacity * d["plane_prefix"]: capacity * d["plane_prefix"] + n * d["plane_stride"]].reshape(n, d["plane_stride"])
            plane[:, :d["nucleotide_cardinality"]] = rows[:, offset: offset + d["nucleotide_cardinality"]]
        return out

    # ---- extraction -----------------------------------------------------------------------------------

    def stage_reads(self, segments: Sequence[Dict[str, np.ndarray]], flags: Optional[np.ndarray], capacity: Optional[int] = None):
        """segments[s] = {"code": uint8[], "quality": uint8[], "offset": int64[n], "length": int32[n]}."""
        n = len(segments[0]["offset"])
        k = len(segments)
        capacity = n if capacity is None else capacity
        base = np.zeros(max(1, capacity * self.plane_width), dtype=np.uint8)
        quality = np.zeros_like(base)
        out_flag = np.zeros(max(1, capacity), dtype=np.uint8)
        kept = np.zeros(max(1, capacity), dtype=np.int64)
        n_kept = ctypes.c_int64(0)
        incoming = ctypes.c_uint64(0)
        incoming_pf = ctypes.c_uint64(0)
        keep = [np.ascontiguousarray(s["code"], dtype=np.uint8) for s in segments], \
               [np.ascontiguousarray(s["quality"], dtype=np.uint8) for s in segments], \
               [np.ascontiguousarray(s["offset"], dtype=np.int64) for s in segments], \
               [np.ascontiguousarray(s["length"], dtype=np.int32) for s in segments]
        arrays = [(ctypes.c_void_p * k)(*[a.ctypes.data for a in group]) for group in keep]
        flags_c = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self.lib.phq_stage_reads(self.handle, n, k, arrays[0], arrays[1], arrays[2], arrays[3], _pointer(flags_c),
                                             capacity, base.ctypes.data, quality.ctypes.data, out_flag.ctypes.data,
                                             kept.ctypes.data, ctypes.byref(n_kept), ctypes.byref(incoming), ctypes.byref(incoming_pf)))
        return {"base": base, "quality": quality, "flag": out_flag, "kept": kept[:n_kept.value], "n": n_kept.value,
                "capacity": capacity, "incoming": (incoming.value, incoming_pf.value)}

    # ---- classification -------------------------------------------------------------------------------

    def classify_device(self, capacity: int, first: int, n: int, base, quality, flag, result, word, stream: int = 0):
        self._check(self.lib.phq_classify_device(self.handle, capacity, first, n, _pointer(base), _pointer(quality),
                                                 _pointer(flag), _pointer(result), _pointer(word), stream or None))

    def classify_host(self, n: int, base: np.ndarray, quality: np.ndarray, flag: Optional[np.ndarray], capacity: Optional[int] = None):
        """Host planes laid out for `capacity` reads (default n); decodes the first n."""
        capacity = n if capacity is None else capacity
        result = np.zeros((max(1, self.n_result), max(1, capacity)), dtype=RESULT_DTYPE)
        word = np.zeros(max(1, capacity), dtype=np.uint32)
        self._check(self.lib.phq_classify_host(self.handle, capacity, n, _pointer(base), _pointer(quality), _pointer(flag),
                                               result.ctypes.data, word.ctypes.data))
        return result[:self.n_result, :n], word[:n]

    def pack_planes(self, n: int, base: np.ndarray, quality: np.ndarray, capacity: Optional[int] = None):
        """phq_pack_planes: (packed planes, escape list) of host planes laid out for `capacity` reads (default n)."""
        capacity = n if capacity is None else capacity
        packed = np.zeros(base.size, dtype=np.uint8)
        room = max(1024, n // 16)
        while True:
            escapes = np.zeros(room, dtype=np.uint64)
            needed = ctypes.c_int64(0)
            self._check(self.lib.phq_pack_planes(self.handle, capacity, n, _pointer(base), _pointer(quality), packed.ctypes.data,
                                                 escapes.ctypes.data, room, ctypes.byref(needed)))
            if needed.value <= room:
                return packed, escapes[:needed.value]
            room = needed.value

    def classify_host_packed(self, n: int, packed: np.ndarray, escapes: np.ndarray, flag: Optional[np.ndarray], capacity: Optional[int] = None,
                             with_word: bool = True):
        """phq_classify_host_packed: classify_host for planes of one byte per nucleotide."""
        capacity = n if capacity is None else capacity
        result = np.zeros((max(1, self.n_result), max(1, capacity)), dtype=RESULT_DTYPE)
        word = np.zeros(max(1, capacity), dtype=np.uint32) if with_word else None
        self._check(self.lib.phq_classify_host_packed(self.handle, capacity, n, _pointer(packed), escapes.ctypes.data if escapes.size else None, int(escapes.size),
                                                      _pointer(flag), result.ctypes.data, word.ctypes.data if with_word else None))
        return result[:self.n_result, :n], (word[:n] if with_word else None)

    def transcode_fastq_to_sam(self, inputs: Sequence[str], output: str, batch_capacity: int = 1 << 20):
        """FASTQ files in (one per input segment), SAM records out; returns the incoming (count, pf count)."""
        paths = (ctypes.c_char_p * len(inputs))(*[p.encode() for p in inputs])
        incoming = ctypes.c_uint64(0)
        incoming_pf = ctypes.c_uint64(0)
        self._check(self.lib.phq_transcode_fastq_to_sam(self.handle, len(inputs), paths, output.encode(), batch_capacity,
                                                        ctypes.byref(incoming), ctypes.byref(incoming_pf)))
        return incoming.value, incoming_pf.value

    def transcode_job(self, batch_capacity: int = 1 << 20):
        """phq_transcode_job: the job's own `input` / `output` directives; returns the incoming (count, pf count)."""
        incoming = ctypes.c_uint64(0)
        incoming_pf = ctypes.c_uint64(0)
        self._check(self.lib.phq_transcode_job(self.handle, batch_capacity, ctypes.byref(incoming), ctypes.byref(incoming_pf)))
        return incoming.value, incoming_pf.value

    def transcode_job_copies(self, copies: Sequence["DecoderSet"], batch_capacity: int = 1 << 20):
        """phq_transcode_job_copies: the job's own run with further copies of this set (one per device) behind the pull loop."""
        incoming = ctypes.c_uint64(0)
        incoming_pf = ctypes.c_uint64(0)
        handles = (ctypes.c_void_p * max(1, len(copies)))(*[c.handle for c in copies])
        self._check(self.lib.phq_transcode_job_copies(self.handle, len(copies), handles, batch_capacity, ctypes.byref(incoming), ctypes.byref(incoming_pf)))
        return incoming.value, incoming_pf.value

    def transcode_plan(self) -> Dict[str, Any]:
        """phq_transcode_plan: the feeds and channel routing phq_transcode_job would use (no device needed)."""
        out = ctypes.c_void_p()
        self._check(self.lib.phq_transcode_plan(self.handle, ctypes.byref(out)))
        try:
            return json.loads(ctypes.string_at(out.value).decode())
        finally:
            self.lib.phq_free_string(out)

    def adjust_prior(self, words: np.ndarray, job: Dict[str, Any]) -> Dict[str, Any]:
        """phq_adjust_prior: `job` with estimated noise / concentration priors from the counters in `words`."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        out = ctypes.c_void_p()
        self._check(self.lib.phq_adjust_prior(self.handle, words.ctypes.data, json.dumps(job).encode(), ctypes.byref(out)))
        try:
            return json.loads(ctypes.string_at(out.value).decode())
        finally:
            self.lib.phq_free_string(out)

    KERNEL_KINDS = ("pamld_product", "pamld_literal", "mdd", "count", "mdd_exact", "pamld_exact", "pamld_near")

    def profile_enable(self, enable: bool = True):
        self._check(self.lib.phq_profile_enable(self.handle, 1 if enable else 0))

    def profile_collect(self) -> Dict[str, Dict[str, float]]:
        """{kernel kind: {"ms": summed event time, "launches": n}} since the last collect."""
        ms = (ctypes.c_double * len(self.KERNEL_KINDS))()
        launches = (ctypes.c_int64 * len(self.KERNEL_KINDS))()
        self._check(self.lib.phq_profile_collect(self.handle, ms, launches))
        return {name: {"ms": ms[k], "launches": launches[k]} for k, name in enumerate(self.KERNEL_KINDS) if launches[k]}

    # ---- statistics -----------------------------------------------------------------------------------

    def profile_chain(self) -> List[Dict[str, Any]]:
        """phq_profile_chain: every timed launch since the last collect, in launch order, under the kernel's name."""
        out = ctypes.c_void_p()
        self._check(self.lib.phq_profile_chain(self.handle, ctypes.byref(out)))
        try:
            return json.loads(ctypes.string_at(out.value).decode())
        finally:
            self.lib.phq_free_string(out)

    def plan_chain(self, n_read: int, mode: int = MODE_AUTO) -> List[Dict[str, Any]]:
        """phq_plan_chain: the kernels a batch of n_read reads would take, named by the stage dispatcher with its launches off."""
        out = ctypes.c_void_p()
        self._check(self.lib.phq_plan_chain(self.handle, n_read, mode, ctypes.byref(out)))
        try:
            return json.loads(ctypes.string_at(out.value).decode())
        finally:
            self.lib.phq_free_string(out)

    def stats_reset(self, stream: int = 0):
        self._check(self.lib.phq_stats_reset(self.handle, stream or None))

    def stats_device_pointer(self) -> int:
        pointer = ctypes.c_void_p()
        self._check(self.lib.phq_stats_device_pointer(self.handle, ctypes.byref(pointer)))
        return pointer.value

    def stats_tensor(self):
        """The device counter buffer as an int64 torch tensor (for torch.distributed.all_reduce over RCCL)."""
        import torch
        return torch.as_tensor(_CudaArray(self.stats_device_pointer(), self.stats_words, self), device="cuda:%d" % self.device)

    def stats_export(self) -> np.ndarray:
        words = np.zeros(self.stats_words, dtype=np.uint64)
        self._check(self.lib.phq_stats_export(self.handle, words.ctypes.data))
        return words

    def stats_import(self, words: np.ndarray):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape == (self.stats_words,)
        self._check(self.lib.phq_stats_import(self.handle, words.ctypes.data))

    def stats_merge(self, other: "DecoderSet"):
        self._check(self.lib.phq_stats_merge(self.handle, other.handle))

    def summarize(self, result: np.ndarray, n_read: Optional[int] = None):
        """Per type accumulators (distance[n, 3] uint32, confidence[n, 3] float64; kind order sample, molecular,
        cellular) of host result planes shaped (n_result, batch_capacity)."""
        result = np.ascontiguousarray(result, dtype=RESULT_DTYPE).reshape(max(1, self.n_result), -1) if self.n_result else np.zeros((1, n_read or 0), dtype=RESULT_DTYPE)
        capacity = result.shape[1]
        n = capacity if n_read is None else n_read
        distance = np.zeros((n, 3), dtype=np.uint32)
        confidence = np.zeros((n, 3), dtype=np.float64)
        self._check(self.lib.phq_summarize(self.handle, capacity, n, result.ctypes.data, distance.ctypes.data, confidence.ctypes.data))
        return distance, confidence

    def qc_export(self) -> np.ndarray:
        """phq_qc_export: the flat image of the quality control accumulators (empty shape when the job has them off)."""
        n = ctypes.c_int64(0)
        self._check(self.lib.phq_qc_export(self.handle, None, 0, ctypes.byref(n)))
        words = np.zeros(max(1, n.value), dtype=np.uint64)
        self._check(self.lib.phq_qc_export(self.handle, words.ctypes.data, n.value, ctypes.byref(n)))
        return words[:n.value]

    def qc_merge(self, words: np.ndarray):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        self._check(self.lib.phq_qc_merge(self.handle, words.ctypes.data, words.size))

    def stats_allreduce(self, rccl_comm: int, stream: int = 0):
        self._check(self.lib.phq_stats_allreduce(self.handle, rccl_comm, stream or None))

    def stats_options(self, words: np.ndarray, decoder: int) -> np.ndarray:
        words = np.ascontiguousarray(words, dtype=np.uint64)
        option = np.zeros(self.decoders[decoder]["n_barcode"] + 1, dtype=OPTION_DTYPE)
        self._check(self.lib.phq_stats_options(self.handle, words.ctypes.data, decoder, option.ctypes.data))
        return option

    def report(self, words: np.ndarray, incoming=(0, 0)) -> str:
        words = np.ascontiguousarray(words, dtype=np.uint64)
        text = ctypes.c_char_p()
        self._check(self.lib.phq_report_json(self.handle, words.ctypes.data, incoming[0], incoming[1], ctypes.byref(text)))
        return text.value.decode()
