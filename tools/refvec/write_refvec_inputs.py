#!/usr/bin/env python3
"""Writes the inputs of tools/refvec/dump_reference_vectors: one binary PGM per image + manifest.txt.

    python tools/refvec/write_refvec_inputs.py OUTDIR

manifest.txt, one case per line:   name width height nfeatures stereo full
(images: OUTDIR/<name>.pgm and, for stereo cases, OUTDIR/<name>_right.pgm).  The pixels are the seeded synthetic frames of
orb_slam2v2-1_amd/synth.py (table: oracle/refvec.py CASES), so the tests regenerate them and compare checksums; nothing from the
reference is involved here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


def main():
    from oracle import refvec
    out = sys.argv[1] if len(sys.argv) > 1 else "refvec_inputs"
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "manifest.txt"), "w") as mf:
        for case in refvec.CASES:
            name, w, h, nf, stereo, seed, kind, full = case
            left, right = refvec.case_images(case)
            write_pgm(os.path.join(out, name + ".pgm"), left)
            if stereo:
                write_pgm(os.path.join(out, name + "_right.pgm"), right)
            mf.write("%s %d %d %d %d %d\n" % (name, w, h, nf, stereo, full))
            print("wrote", name, left.shape)
    print("manifest:", os.path.join(out, "manifest.txt"))


if __name__ == "__main__":
    main()
