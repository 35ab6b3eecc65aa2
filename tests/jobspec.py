"""Job specification helpers: the subset of the reference's configuration *compiler* that the
classification path depends on.

The product boundary (include/pheniqs_amd.h) consumes an already **compiled** job document — the
shape `pheniqs mux --compile` prints (reference test/BDGGG/valid/compile_annotated.out).  This module
produces that shape from a terse user-level spec so that tests and bench.py can build codecs without
the reference binary.  It restates, for the decoder topics only:

* Transcode::compile_transformation          transcode.cpp:735-763   (default knit = one segment per token)
* Transcode::compile_decoder_transformation  transcode.cpp:824-940   (cardinalities, random barcode probability,
                                                                      undetermined "====", length / duplicate checks)
* Transcode::compile_decoder                 transcode.cpp:941-1039  (index, concentration * (1-noise)/sum, BC)
* CodecMetric::compile_barcode_tolerance     metric.h:87-111,216-242 (shannon bound, default distance tolerance)
* projections "<type>:decoder"/"<type>:barcode" configuration.json:423-520 (restated as DEFAULTS below)

Import / base inheritance, IO, SAM header atoms and the CLI are out of scope.
"""
from __future__ import annotations

import copy
import re
from typing import Any, Dict, List, Optional

TOPICS = ("sample", "molecular", "cellular")

# configuration.json:423-520 ("cellular:decoder", "molecular:decoder", "sample:decoder") and :368-376
DECODER_DEFAULTS: Dict[str, Dict[str, Any]] = {
    "sample": {"algorithm": "pamld"},
    "cellular": {"algorithm": "pamld"},
    "molecular": {"algorithm": "naive"},
}
COMMON_DEFAULTS: Dict[str, Any] = {
    "confidence threshold": 0.95,
    "high quality distance threshold": 0,
    "high quality threshold": 30,
    "noise": 0.01,
    "quality masking threshold": 0,
    "corrected quality": 30,
}
BARCODE_DEFAULT_CONCENTRATION = 1
# configuration.json:419-490 "sample:decoder" / "sample:barcode": read group members the sample decoder takes from the
# job's root and hands to its barcodes (Transcode::compile_topic transcode.cpp:769-800)
SAMPLE_PROJECTED = ("CN", "DT", "LB", "PG", "PI", "PL", "PM", "SM", "flowcell id", "flowcell lane number")

_TOKEN_RE = re.compile(r"^(s|c|m|[0-9]+):(-?[0-9]+)?:(-?[0-9]+)?$")   # transform.cpp:31


class ConfigurationError(ValueError):
    """error.h:78-86 ConfigurationError (ErrorCode 3)."""


class Token:
    """transform.h:35-88 Token: `segment:start:end` with Python-like negative indices."""

    def __init__(self, pattern: str, index: int):
        match = _TOKEN_RE.match(pattern)
        if not match:
            raise ConfigurationError("illegal token syntax " + pattern)
        head = match.group(1)
        self.index = index
        self.input_segment_index = {"s": -1, "c": -2, "m": -3}.get(head, None)
        if self.input_segment_index is None:
            self.input_segment_index = int(head)
        self.start = int(match.group(2)) if match.group(2) else 0
        self.end_terminated = bool(match.group(3))
        self.end = int(match.group(3)) if match.group(3) else 0

    def constant(self) -> bool:
        if self.end_terminated:
            return (self.start >= 0 and self.end >= 0) or (self.start < 0 and self.end < 0)
        return self.start < 0

    def empty(self) -> bool:
        return (self.end_terminated and self.start >= self.end) and (
            (self.start >= 0 and self.end >= 0) or (self.start < 0 and self.end < 0))

    def length(self) -> int:
        if self.constant():
            if self.end_terminated:
                return 0 if self.empty() else self.end - self.start
            return -self.start
        return -1


def parse_rule(transform: Dict[str, Any]) -> List[Dict[str, int]]:
    """decode_value_by_key<Rule> transform.cpp:252-331: flat list of transforms in knit order."""
    tokens = [Token(pattern, i) for i, pattern in enumerate(transform["token"])]
    result = []
    for output_segment_index, pattern in enumerate(transform["knit"]):
        for piece in pattern.split(":"):
            reverse_complement = piece.startswith("~")
            body = piece[1:] if reverse_complement else piece
            if not body.isdigit():
                raise ConfigurationError("illegal character in transform " + pattern)
            token_index = int(body)
            if token_index >= len(tokens):
                raise ConfigurationError("invalid token reference %d in transform" % token_index)
            token = tokens[token_index]
            result.append({
                "input_segment_index": token.input_segment_index,
                "start": token.start,
                "end": token.end,
                "end_terminated": int(token.end_terminated),
                "output_segment_index": output_segment_index,
                "reverse_complement": int(reverse_complement),
                "length": token.length(),
            })
    return result


def _shannon_bound(words: List[str], barcode_length: int) -> int:
    """WordMetric::find_shannon_bound metric.h:87-111 over the *set* of distinct segment words."""
    words = sorted(set(words))
    min_distance = barcode_length
    for i in range(len(words)):
        for j in range(i + 1, len(words)):
            distance = sum(1 for a, b in zip(words[i], words[j]) if a != b)
            if distance < min_distance:
                min_distance = distance
    return int((min_distance - 1) / 2)    # C integer division truncates toward zero


def _infer_PU_and_ID(barcode: Dict[str, Any], undetermined: bool) -> None:
    """Transcode::infer_PU / infer_ID transcode.cpp:1224-1260."""
    if not isinstance(barcode.get("PU"), str):
        suffix = "undetermined" if undetermined else "".join(barcode.get("barcode") or [])
        if suffix:
            unit = ""
            if isinstance(barcode.get("flowcell id"), str):
                unit = barcode["flowcell id"] + ":"
                if isinstance(barcode.get("flowcell lane number"), int):
                    unit += "%d:" % barcode["flowcell lane number"]
            barcode["PU"] = unit + suffix
    if not isinstance(barcode.get("ID"), str) and isinstance(barcode.get("PU"), str):
        barcode["ID"] = barcode["PU"]


def compile_decoder(spec: Dict[str, Any], kind: str, index: int, input_segment_cardinality: Optional[int] = None,
                    root: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    value = copy.deepcopy(spec)
    root = root or {}
    value["index"] = index
    projected = SAMPLE_PROJECTED if kind == "sample" else ()
    for key in list(COMMON_DEFAULTS) + list(DECODER_DEFAULTS[kind]) + list(projected):
        if value.get(key) is None and root.get(key) is not None:
            value[key] = copy.deepcopy(root[key])
    for key, default in {**COMMON_DEFAULTS, **DECODER_DEFAULTS[kind]}.items():
        value.setdefault(key, default)
    algorithm = value["algorithm"]
    barcode_default = {key: copy.deepcopy(value.get(key, root.get(key))) for key in projected
                       if value.get(key, root.get(key)) is not None}

    # transcode.cpp:735-763
    transform = value.get("transform")
    if transform is None:
        raise ConfigurationError("decoder is missing a transform")
    if not transform.get("knit"):
        transform["knit"] = [str(i) for i in range(len(transform["token"]))]

    # transcode.cpp:824-863
    rule = parse_rule(transform)
    segment_cardinality = len(transform["knit"])
    for token_index, pattern in enumerate(transform["token"]):
        token = Token(pattern, token_index)
        if input_segment_cardinality is not None and not token.input_segment_index < input_segment_cardinality:
            raise ConfigurationError("invalid input feed reference %d in token %d" % (token.input_segment_index, token_index))
        if token.empty():
            raise ConfigurationError("token %s is empty" % pattern)
        if not token.constant():
            raise ConfigurationError("token %s is not fixed width" % pattern)
    barcode_length = [0] * segment_cardinality
    nucleotide_cardinality = 0
    for t in rule:
        barcode_length[t["output_segment_index"]] += t["length"]
        nucleotide_cardinality += t["length"]
    value["segment cardinality"] = segment_cardinality
    value["nucleotide cardinality"] = nucleotide_cardinality
    value["barcode length"] = barcode_length
    lower_bound = 1.0 / float(pow(4, nucleotide_cardinality))
    if "random barcode probability" in value:
        if value["random barcode probability"] < lower_bound:
            raise ConfigurationError("random barcode probability is smaller than lower bound")
    else:
        value["random barcode probability"] = lower_bound

    # transcode.cpp:867-882, 974-986
    noise = float(value["noise"])
    undetermined = value.setdefault("undetermined", {})
    for key, default in barcode_default.items():
        undetermined.setdefault(key, copy.deepcopy(default))
    undetermined["barcode"] = ["=" * n for n in barcode_length]
    undetermined["segment cardinality"] = segment_cardinality
    undetermined["index"] = 0
    undetermined["algorithm"] = algorithm
    _infer_PU_and_ID(undetermined, True)
    undetermined["concentration"] = noise

    codec = value.get("codec")
    if codec:
        # sort_json_value json.cpp:875-893: members are ordered by key before compile_decoder walks them
        ordered = {key: codec[key] for key in sorted(codec)}
        seen = set()
        unique_id = {undetermined.get("ID")}
        barcode_index = 1
        total_concentration = 0.0
        for key, record in ordered.items():
            segments = record.get("barcode")
            if segments is None or len(segments) != segment_cardinality:
                raise ConfigurationError("expected %d segments in barcode %s" % (segment_cardinality, key))
            for segment_index, segment in enumerate(segments):
                if len(segment) != barcode_length[segment_index]:
                    raise ConfigurationError("expected %d but found %d nucleotides in segment %d of barcode %s" % (
                        barcode_length[segment_index], len(segment), segment_index, key))
            flat = "".join(segments)
            if flat in seen:
                raise ConfigurationError("duplicate barcode sequence " + flat)
            seen.add(flat)
            for default_key, default in barcode_default.items():
                record.setdefault(default_key, copy.deepcopy(default))
            record["segment cardinality"] = segment_cardinality
            record["algorithm"] = algorithm
            record["index"] = barcode_index
            _infer_PU_and_ID(record, False)
            if record.get("ID") in unique_id:
                raise ConfigurationError("duplicate %s barcode" % key)
            unique_id.add(record.get("ID"))
            record.setdefault("concentration", BARCODE_DEFAULT_CONCENTRATION)
            concentration = float(record["concentration"])
            if concentration < 0:
                raise ConfigurationError("barcode concentration must be a positive number")
            total_concentration += concentration
            record["BC"] = "-".join(segments)
            barcode_index += 1
        value["barcode cardinality"] = barcode_index
        if not total_concentration > 0:
            raise ConfigurationError("total pool concentration is not a positive number")
        factor = (1.0 - noise) / total_concentration          # transcode.cpp:1024-1029
        for record in ordered.values():
            record["concentration"] = float(record["concentration"]) * factor
        value["codec"] = ordered

        # metric.h:216-242
        shannon_bound = []
        for segment_index in range(segment_cardinality):
            words = [record["barcode"][segment_index] for record in ordered.values()]
            shannon_bound.append(_shannon_bound(words, barcode_length[segment_index]))
        value["shannon bound"] = shannon_bound
        if "distance tolerance" in value and value["distance tolerance"] is not None:
            tolerance = value["distance tolerance"]
            if len(tolerance) != segment_cardinality:
                raise ConfigurationError("distance tolerance cardinality inconsistant with barcode segment cardinality")
            for segment_index, t in enumerate(tolerance):
                if t > shannon_bound[segment_index]:
                    raise ConfigurationError("barcode tolerance for segment %d is higher than shannon bound %d" % (
                        segment_index, shannon_bound[segment_index]))
        else:
            value["distance tolerance"] = list(shannon_bound)
    return value


def compile_job(spec: Dict[str, Any]) -> Dict[str, Any]:
    """Compile the decoder topics of a job spec.  Keys other than the topics are copied through."""
    job = {key: copy.deepcopy(v) for key, v in spec.items() if key not in TOPICS}
    cardinality = job.get("input segment cardinality")
    for kind in TOPICS:
        if kind not in spec or spec[kind] is None:
            continue
        node = spec[kind]
        if isinstance(node, dict):
            job[kind] = compile_decoder(node, kind, 0, cardinality, job)
        else:
            job[kind] = [compile_decoder(element, kind, i, cardinality, job) for i, element in enumerate(node)]
    # Transcode::find_multiplexing_decoder transcode.cpp:1087-1223, reduced to its common outcome:
    # an explicitly marked decoder wins, otherwise the sample decoder drives output routing.
    marked = [d for d in iter_decoders(job) if d[1].get("multiplexing classifier")]
    if not marked and isinstance(job.get("sample"), dict):
        job["sample"]["multiplexing classifier"] = True
    job.setdefault("filter incoming qc fail", False)
    job.setdefault("leading segment index", 0)
    return job


def iter_decoders(job: Dict[str, Any]):
    """(kind, decoder) in the order TranscodingDecoder::classify runs them: transcode.h:51-60."""
    for kind in TOPICS:
        node = job.get(kind)
        if node is None:
            continue
        if isinstance(node, dict):
            yield kind, node
        else:
            for element in node:
                yield kind, element
