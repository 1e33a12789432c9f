#!/usr/bin/env python3
"""Extracts the NUMERIC LITERALS of the reference's hot path -- numbers only, no source text -- into
tests/golden/reference_literals.json, so that tests/test_golden.py can check that the CPU oracle (oracle/rl_oracle.cpp) and
the product (csrc/rl_core.h, rl_kernels.hip.h, rl_scene.cpp) carry exactly these values: the reference holds no
numeric test vectors (main.rs:69-74), so this pins what CAN be pinned to reference-held data -- every constant a
restatement could mistype -- the way tools/gen_cie_table.py pins the CIE tables.
Run in the build container only (needs /root/reference).  Output: {"<file>:<first>-<last>": ["<literal>", ...]} in source
order, literals as written (suffixes like f32 / u32 / isize dropped)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src"
RANGES = [
    ("srgb.rs", 20, 33),            # gamma curve and the XYZ -> sRGB matrix
    ("material.rs", 61, 74),        # Planck's law
    ("material.rs", 93, 99),        # black-body normalisation (Wien peak, nm)
    ("material.rs", 155, 160),      # coloured diffuse: gaussian reflectance
    ("material.rs", 203, 213),      # SF10 Sellmeier coefficients
    ("material.rs", 267, 298),      # soap bubble: reflection threshold, phase, clamp, weights
    ("constants.rs", 17, 25),       # golden ratio, h, k, c, Wien
    ("trace_unit.rs", 66, 67),      # photons per batch
    ("trace_unit.rs", 84, 88),      # initial continue chance / intensity
    ("trace_unit.rs", 114, 123),    # origin offset, decay, Russian roulette
    ("camera.rs", 56, 56),          # screen distance
    ("camera.rs", 95, 102),         # depth of field, chromatic zoom
    ("monte_carlo.rs", 25, 58),     # ranges of the random quantities, hemisphere sample
    ("vector3.rs", 69, 83),         # rotate_towards thresholds
    ("scene.rs", 42, 42),           # "infinite" distance
    ("cie1931.rs", 20, 48),         # table lookup
    ("plot_unit.rs", 56, 84),       # splat coordinates
    ("tonemap_unit.rs", 55, 100),   # exposure, log curve, quantisation
    ("geometry.rs", 204, 240),      # sphere quadratic
    ("geometry.rs", 298, 341),      # paraboloid quadratic
    ("geometry.rs", 421, 436),      # infinite prism: in-radius, vertex angles
    ("app.rs", 172, 357),           # the scene and the camera
]
NUMBER = re.compile(r"(?<![A-Za-z_0-9.])(\d+\.\d*(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|\d+)(?:_?(?:f32|f64|u32|u64|usize|isize|i32|i64))?(?![A-Za-z_0-9])")


def literals(path, first, last):
    out = []
    for line in open(os.path.join(SRC, path)).read().split("\n")[first - 1:last]:
        line = line.split("//")[0]                      # comments hold prose numbers
        line = re.sub(r'"[^"]*"', "", line)             # so do string literals
        line = re.sub(r"\b(?:sp|p|a|sky|wall|surface|material|t|ab|x|y|z|c|q)\d+\b", "", line)   # identifiers that end in digits
        line = re.sub(r"\.(\d+)\b(?=\s*[^\d.eE])", lambda m: "." + m.group(1), line)
        out += [m.group(1) for m in NUMBER.finditer(line)]
    return out


def main():
    table = {"%s:%d-%d" % (p, a, b): literals(p, a, b) for p, a, b in RANGES}
    with open(os.path.join(ROOT, "tests", "golden", "reference_literals.json"), "w") as f:
        json.dump(table, f, indent=0)
    for k, v in table.items():
        print(k, v)


if __name__ == "__main__":
    main()
