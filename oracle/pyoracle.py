"""ctypes front end of the CPU oracle (oracle/pheniqs_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (pheniqs_amd/) never does.  The compiled job document is flattened here independently of the
product's C++ JSON reader, so the two config paths cross-check each other.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY = os.path.join(HERE, "libpheniqs_oracle.so")

MAX_SEGMENT = 8
MAX_TRANSFORM = 16
ALGORITHM = {"passthrough": 0, "pamld": 1, "mdd": 2, "naive": 3}
KIND = {"sample": 0, "molecular": 1, "cellular": 2}
TOPICS = ("sample", "molecular", "cellular")

RESULT_DTYPE = np.dtype([("index", "<i4"), ("distance", "u1"), ("hq_distance", "u1"), ("status", "u1"),
                         ("qcfail", "u1"), ("confidence", "<f8")])
OPTION_DTYPE = np.dtype([("count", "<u8"), ("pf_count", "<u8"), ("accumulated_distance", "<u8"),
                         ("accumulated_pf_distance", "<u8"), ("low_conditional_confidence_count", "<u8"),
                         ("low_confidence_count", "<u8"), ("accumulated_confidence", "<f8"),
                         ("accumulated_pf_confidence", "<f8")])
OPTION_FINAL_FIELDS = ("pf_fraction", "average_distance", "average_confidence", "average_pf_distance",
                       "average_pf_confidence", "pooled_fraction", "pf_pooled_fraction",
                       "pooled_classified_fraction", "pf_pooled_classified_fraction",
                       "estimated_concentration_prior")
OPTION_FINAL_DTYPE = np.dtype([(name, "<f8") for name in OPTION_FINAL_FIELDS])


class Transform(ctypes.Structure):
    _fields_ = [("input_segment_index", ctypes.c_int32), ("start", ctypes.c_int32), ("end", ctypes.c_int32),
                ("end_terminated", ctypes.c_int32), ("output_segment_index", ctypes.c_int32),
                ("reverse_complement", ctypes.c_int32)]


class Decoder(ctypes.Structure):
    _fields_ = [("algorithm", ctypes.c_int32), ("kind", ctypes.c_int32), ("n_barcode", ctypes.c_int32),
                ("n_segment", ctypes.c_int32), ("segment_length", ctypes.c_int32 * MAX_SEGMENT),
                ("nucleotide_cardinality", ctypes.c_int32), ("multiplexing_classifier", ctypes.c_int32),
                ("high_quality_threshold", ctypes.c_int32), ("high_quality_distance_threshold", ctypes.c_int32),
                ("quality_masking_threshold", ctypes.c_int32), ("distance_tolerance", ctypes.c_int32 * MAX_SEGMENT),
                ("n_transform", ctypes.c_int32), ("transform", Transform * MAX_TRANSFORM),
                ("noise", ctypes.c_double), ("confidence_threshold", ctypes.c_double),
                ("random_barcode_probability", ctypes.c_double),
                ("barcode", ctypes.c_void_p), ("concentration", ctypes.c_void_p), ("barcode_index", ctypes.c_void_p)]


class SelectorFinal(ctypes.Structure):
    _fields_ = [(name, ctypes.c_uint64) for name in (
        "count", "pf_count", "classified_count", "pf_classified_count", "accumulated_classified_distance",
        "accumulated_pf_classified_distance", "low_conditional_confidence_count", "low_confidence_count")] + [
        (name, ctypes.c_double) for name in (
        "accumulated_classified_confidence", "accumulated_pf_classified_confidence", "pf_fraction",
        "classified_fraction", "average_classified_distance", "average_classified_confidence",
        "pf_classified_fraction", "classified_pf_fraction", "average_pf_classified_distance",
        "average_pf_classified_confidence", "estimated_noise_prior")]


_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    sources = [os.path.join(HERE, "pheniqs_oracle.c"), os.path.join(HERE, "pheniqs_oracle.h")]
    stale = force or not os.path.exists(LIBRARY) or any(os.path.getmtime(s) > os.path.getmtime(LIBRARY) for s in sources)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "libpheniqs_oracle.so"])
    return LIBRARY


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIBRARY)
        L.orc_substitution_quality.restype = ctypes.c_double
        L.orc_substitution_quality.argtypes = [ctypes.c_uint8] * 3
        L.orc_true_positive_quality.restype = ctypes.c_double
        L.orc_true_positive_quality.argtypes = [ctypes.c_uint8]
        L.orc_uniform_base_quality.restype = ctypes.c_double
        L.orc_phred_probability_base.restype = ctypes.c_double
        L.orc_ascii_to_bam.restype = ctypes.c_uint8
        L.orc_ascii_to_bam.argtypes = [ctypes.c_uint8]
        L.orc_reverse_complement_bam.restype = ctypes.c_uint8
        L.orc_reverse_complement_bam.argtypes = [ctypes.c_uint8]
        L.orc_job_create.restype = ctypes.c_void_p
        L.orc_job_create.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.orc_job_destroy.argtypes = [ctypes.c_void_p]
        L.orc_job_reset.argtypes = [ctypes.c_void_p]
        L.orc_job_row_width.argtypes = [ctypes.c_void_p]
        L.orc_job_result_count.argtypes = [ctypes.c_void_p]
        L.orc_job_decoder_offset.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.orc_stage_read.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_classify.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5
        L.orc_classify_summary.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 7
        L.orc_job_option.restype = ctypes.c_void_p
        L.orc_job_option.argtypes = [ctypes.c_void_p, ctypes.c_int32]
        L.orc_job_outgoing.argtypes = [ctypes.c_void_p] * 3
        L.orc_job_collect.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_job_finalize.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_compensated_decoding_probability.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4
        L.orc_phred_init()
        _lib = L
    return _lib


def ascii_to_bam(text: bytes) -> np.ndarray:
    table = np.array([lib().orc_ascii_to_bam(c) for c in range(256)], dtype=np.uint8)
    return table[np.frombuffer(text, dtype=np.uint8)]


_TOKEN_HEAD = {"s": -1, "c": -2, "m": -3}


def _parse_transforms(transform: Dict[str, Any]) -> List[Dict[str, int]]:
    """Independent re-reading of token/knit (transform.cpp:118-174,252-331)."""
    tokens = []
    for pattern in transform["token"]:
        head, start, end = pattern.split(":")
        tokens.append({
            "input_segment_index": _TOKEN_HEAD[head] if head in _TOKEN_HEAD else int(head),
            "start": int(start) if start else 0,
            "end": int(end) if end else 0,
            "end_terminated": 1 if end else 0,
        })
    knit = transform.get("knit") or [str(i) for i in range(len(tokens))]
    out = []
    for output_segment_index, pattern in enumerate(knit):
        for piece in pattern.split(":"):
            rc = piece.startswith("~")
            token = dict(tokens[int(piece[1:] if rc else piece)])
            token["output_segment_index"] = output_segment_index
            token["reverse_complement"] = int(rc)
            out.append(token)
    return out


def job_decoders(job: Dict[str, Any]):
    for kind in TOPICS:
        node = job.get(kind)
        if node is None:
            continue
        for element in ([node] if isinstance(node, dict) else node):
            yield kind, element


class OracleJob:
    """One TranscodingDecoder worth of decoders (transcode.h:40-77) on the CPU oracle."""

    def __init__(self, job: Dict[str, Any]):
        self.lib = lib()
        self.decoders = list(job_decoders(job))
        self._keep = []
        array = (Decoder * max(1, len(self.decoders)))()
        for i, (kind, d) in enumerate(self.decoders):
            c = array[i]
            c.algorithm = ALGORITHM[d.get("algorithm", "passthrough")]
            c.kind = KIND[kind]
            codec = d.get("codec") or {}
            records = list(codec.values())
            c.n_barcode = len(records)
            c.n_segment = int(d["segment cardinality"])
            lengths = list(d["barcode length"])
            for k, n in enumerate(lengths):
                c.segment_length[k] = int(n)
            c.nucleotide_cardinality = int(d["nucleotide cardinality"])
            c.multiplexing_classifier = 1 if d.get("multiplexing classifier") else 0
            c.high_quality_threshold = int(d.get("high quality threshold", 30))
            c.high_quality_distance_threshold = int(d.get("high quality distance threshold", 0))
            c.quality_masking_threshold = int(d.get("quality masking threshold", 0))
            for k, t in enumerate(d.get("distance tolerance") or []):
                c.distance_tolerance[k] = int(t)
            transforms = _parse_transforms(d["transform"])
            c.n_transform = len(transforms)
            for k, t in enumerate(transforms):
                for name, v in t.items():
                    setattr(c.transform[k], name, v)
            c.noise = float(d.get("noise", 0))
            c.confidence_threshold = float(d.get("confidence threshold", 0))
            c.random_barcode_probability = float(d.get("random barcode probability", 0))
            L = c.nucleotide_cardinality
            barcode = np.zeros((max(1, len(records)), max(1, L)), dtype=np.uint8)
            for b, record in enumerate(records):
                barcode[b, :L] = ascii_to_bam("".join(record["barcode"]).encode())
            concentration = np.array([float(r.get("concentration", 0)) for r in records] or [0.0], dtype=np.float64)
            index = np.array([int(r["index"]) for r in records] or [0], dtype=np.int32)
            self._keep += [barcode, concentration, index]
            c.barcode = barcode.ctypes.data
            c.concentration = concentration.ctypes.data
            c.barcode_index = index.ctypes.data
        self._array = array
        self.handle = self.lib.orc_job_create(ctypes.addressof(array), len(self.decoders))
        self.row_width = self.lib.orc_job_row_width(self.handle)
        self.n_result = self.lib.orc_job_result_count(self.handle)
        self.n_barcode = [array[i].n_barcode for i in range(len(self.decoders))]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.orc_job_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def reset(self):
        self.lib.orc_job_reset(self.handle)

    def decoder_offset(self, i: int) -> int:
        return self.lib.orc_job_decoder_offset(self.handle, i)

    def stage(self, reads: Sequence[Sequence[tuple]]):
        """reads[i] = [(code uint8[], quality uint8[]), ...] one tuple per input segment -> (base, qual) rows."""
        n = len(reads)
        base = np.zeros((n, self.row_width), dtype=np.uint8)
        qual = np.zeros((n, self.row_width), dtype=np.uint8)
        for i, segments in enumerate(reads):
            k = len(segments)
            codes = [np.ascontiguousarray(s[0], dtype=np.uint8) for s in segments]
            quals = [np.ascontiguousarray(s[1], dtype=np.uint8) for s in segments]
            pc = (ctypes.c_void_p * k)(*[c.ctypes.data for c in codes])
            pq = (ctypes.c_void_p * k)(*[q.ctypes.data for q in quals])
            pl = (ctypes.c_int32 * k)(*[len(c) for c in codes])
            self.lib.orc_stage_read(self.handle, pc, pq, pl, k, base[i].ctypes.data, qual[i].ctypes.data)
        return base, qual

    def classify(self, base: np.ndarray, qual: np.ndarray, flags: Optional[np.ndarray] = None, summary: bool = False):
        base = np.ascontiguousarray(base, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        n = base.shape[0]
        assert base.shape == (n, self.row_width) and qual.shape == base.shape
        if flags is None:
            flags = np.zeros(n, dtype=np.uint8)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        result = np.zeros((n, max(1, self.n_result)), dtype=RESULT_DTYPE)
        word = np.zeros(n, dtype=np.uint32)
        if summary:
            distance = np.zeros((n, 3), dtype=np.uint32)
            confidence = np.zeros((n, 3), dtype=np.float64)
            self.lib.orc_classify_summary(self.handle, n, base.ctypes.data, qual.ctypes.data, flags.ctypes.data,
                                          result.ctypes.data if self.n_result else None, word.ctypes.data,
                                          distance.ctypes.data, confidence.ctypes.data)
            return result[:, :self.n_result], word, distance, confidence
        self.lib.orc_classify(self.handle, n, base.ctypes.data, qual.ctypes.data, flags.ctypes.data,
                              result.ctypes.data if self.n_result else None, word.ctypes.data)
        return result[:, :self.n_result], word

    def options(self, decoder: int) -> np.ndarray:
        n = self.n_barcode[decoder] + 1
        pointer = self.lib.orc_job_option(self.handle, decoder)
        buffer = (ctypes.c_char * (OPTION_DTYPE.itemsize * n)).from_address(pointer)
        return np.frombuffer(buffer, dtype=OPTION_DTYPE, count=n).copy()

    def outgoing(self):
        count = ctypes.c_uint64(0)
        pf_count = ctypes.c_uint64(0)
        self.lib.orc_job_outgoing(self.handle, ctypes.byref(count), ctypes.byref(pf_count))
        return count.value, pf_count.value

    def collect(self, other: "OracleJob"):
        self.lib.orc_job_collect(self.handle, other.handle)

    def finalize(self, decoder: int):
        selector = SelectorFinal()
        option = np.zeros(self.n_barcode[decoder] + 1, dtype=OPTION_FINAL_DTYPE)
        self.lib.orc_job_finalize(self.handle, decoder, ctypes.byref(selector), option.ctypes.data)
        return {name: getattr(selector, name) for name, _ in SelectorFinal._fields_}, option

    # ---- report (selector.cpp:102-135,215-247; classifier.h:161-177; barcode.cpp:55-69) -------------------

    def report(self, decoder: int) -> Dict[str, Any]:
        kind, d = self.decoders[decoder]
        selector, option = self.finalize(decoder)
        raw = self.options(decoder)
        out: Dict[str, Any] = {"index": int(d["index"]), "count": selector["count"], "pf count": selector["pf_count"],
                               "classified count": selector["classified_count"]}
        if selector["low_conditional_confidence_count"] > 0:
            out["low conditional confidence count"] = selector["low_conditional_confidence_count"]
        if selector["low_confidence_count"] > 0:
            out["low confidence count"] = selector["low_confidence_count"]
        out["pf classified count"] = selector["pf_classified_count"]
        out["pf fraction"] = selector["pf_fraction"]
        out["classified fraction"] = selector["classified_fraction"]
        for key, name in (("average classified distance", "average_classified_distance"),
                          ("average classified confidence", "average_classified_confidence")):
            if selector[name] > 0:
                out[key] = selector[name]
        out["pf classified fraction"] = selector["pf_classified_fraction"]
        out["classified pf fraction"] = selector["classified_pf_fraction"]
        for key, name in (("average pf classified distance", "average_pf_classified_distance"),
                          ("average pf classified confidence", "average_pf_classified_confidence"),
                          ("estimated noise", "estimated_noise_prior")):
            if selector[name] > 0:
                out[key] = selector[name]

        def encode_option(b: int, record: Optional[Dict[str, Any]]) -> Dict[str, Any]:
            o, f = raw[b], option[b]
            e: Dict[str, Any] = {"count": int(o["count"])}
            if f["average_distance"] > 0: e["average distance"] = float(f["average_distance"])
            if f["average_confidence"] > 0: e["average confidence"] = float(f["average_confidence"])
            if o["low_conditional_confidence_count"] > 0: e["low conditional confidence count"] = int(o["low_conditional_confidence_count"])
            if o["low_confidence_count"] > 0: e["low confidence count"] = int(o["low_confidence_count"])
            e["pooled fraction"] = float(f["pooled_fraction"])
            if f["pooled_classified_fraction"] > 0: e["pooled classified fraction"] = float(f["pooled_classified_fraction"])
            e["pf count"] = int(o["pf_count"])
            if f["average_pf_distance"] > 0: e["average pf distance"] = float(f["average_pf_distance"])
            if f["average_pf_confidence"] > 0: e["average pf confidence"] = float(f["average_pf_confidence"])
            e["pf fraction"] = float(f["pf_fraction"])
            e["pf pooled fraction"] = float(f["pf_pooled_fraction"])
            if f["pf_pooled_classified_fraction"] > 0: e["pf pooled classified fraction"] = float(f["pf_pooled_classified_fraction"])
            if f["estimated_concentration_prior"] > 0: e["estimated concentration"] = float(f["estimated_concentration_prior"])
            e["index"] = int(record["index"]) if record is not None else 0
            if record is not None:
                e["concentration"] = float(record["concentration"])
                e["barcode"] = list(record["barcode"])
            return e

        out["unclassified"] = encode_option(0, None)
        records = list((d.get("codec") or {}).values())
        if records:
            out["classified"] = [encode_option(b + 1, r) for b, r in enumerate(records)]
        return out
