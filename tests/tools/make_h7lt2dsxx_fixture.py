#!/usr/bin/env python3
"""Writes tests/golden/H7LT2DSXX/l01_sample_decoder.json from the reference's example job documents (data, not source):
the 94 dual index pairs of lane 1 of flowcell H7LT2DSXX with the priors its prior estimation pass left
(example/H7LT2DSXX/H7LT2DSXX_l01_adjusted.json) and the decoder parameters of the prior estimation run the reference
publishes a wall clock for (example/H7LT2DSXX/H7LT2DSXX_l01_estimate.json, docs/illumina_api_vignette.md:222).
Run in the build container (it reads /root/reference); the GPU box only sees the fixture."""
import json
import os
import sys

REFERENCE = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
adjusted = json.load(open(os.path.join(REFERENCE, "example/H7LT2DSXX/H7LT2DSXX_l01_adjusted.json")))
estimate = json.load(open(os.path.join(REFERENCE, "example/H7LT2DSXX/H7LT2DSXX_l01_estimate.json")))
report = json.load(open(os.path.join(REFERENCE, "example/H7LT2DSXX/H7LT2DSXX_l01_estimate_report.json")))
sample = adjusted["sample"]
out = {
    "source": ["example/H7LT2DSXX/H7LT2DSXX_l01_adjusted.json", "example/H7LT2DSXX/H7LT2DSXX_l01_estimate.json",
               "example/H7LT2DSXX/H7LT2DSXX_l01_estimate_report.json", "docs/illumina_api_vignette.md:222"],
    "published": {"wall clock": "about 1:45 hours per lane (prior estimation run, index reads only, output /dev/null)",
                  "hardware": "dual socket Intel Xeon E5-2620, 12 cores", "reads": report["incoming"]["count"],
                  "derived Mreads/s (whole process)": report["incoming"]["count"] / 6300.0 / 1e6},
    "estimate": {"algorithm": estimate["sample"]["algorithm"], "noise": estimate["sample"]["noise"],
                 "confidence threshold": estimate["sample"]["confidence threshold"], "token": estimate["sample"]["transform"]["token"]},
    "adjusted": {"algorithm": sample["algorithm"], "noise": sample["noise"], "confidence threshold": sample["confidence threshold"],
                 "token": sample["transform"]["token"]},
    "codec": {key: {"barcode": record["barcode"], "concentration": record["concentration"]} for key, record in sample["codec"].items()},
}
path = os.path.join(HERE, "golden", "H7LT2DSXX", "l01_sample_decoder.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(out["codec"]), "barcodes")
