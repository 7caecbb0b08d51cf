#!/usr/bin/env python
"""Builds tests/golden/ref_icons.json: the decoded pixels of a few of the reference's own icon files
(tests/mapcss/symbols/*.png — data files of its test suite) as straight-alpha RGBA8, i.e. what Icon::load
(src/draw/icon.rs:14-58) hands to RgbaColor::from_components.  Used by the image-fill (fill.rs:36-40; the
stylesheet's fill-image rules, mapnik.mapcss:60-306) and label-icon (labeler.rs:91-106) parity tests, SURVEY.md
§8(f) N4.  Run in the build container only (reads /root/reference)."""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SYM = "/root/reference/tests/mapcss/symbols"
NAMES = ["forest.png", "scrub.png", "grave_yard.png", "military_red_hz2.png", "cafe.p.16.png", "station.png", "orchard.png"]


def main():
    out = {"_provenance": __doc__}
    for n in NAMES:
        im = Image.open(os.path.join(SYM, n))
        mode = im.mode
        out[n] = {"mode": mode, "rgba": np.array(im.convert("RGBA")).tolist()}
    with open(os.path.join(HERE, "ref_icons.json"), "w") as f:
        json.dump(out, f)
    print({n: (len(out[n]["rgba"]), len(out[n]["rgba"][0])) for n in NAMES})


if __name__ == "__main__":
    main()
