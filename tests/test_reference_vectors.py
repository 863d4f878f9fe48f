"""Reference-pinned vectors: the ingest path.

The reference ships no tests and no golden vectors, and neither it nor OpenCV can be built in this image, so the CPU oracle is
"parity unpinned" (DESIGN.md section 3).  What CAN exist is vectors a maintainer dumps from a real ORB-SLAM2 + OpenCV build with
tools/refvec/dump_reference_vectors.cc (tests/golden/README.md): files tests/golden/ref_<case>.orbvec.  When they are present,

  * test_reference_vectors_pin_the_oracle      (CPU)   checks the ORACLE against them, stage by stage, and
  * test_reference_vectors_pin_the_hip_path    (-m gpu) checks the HIP path against them;

when they are absent both SKIP with that reason - nothing here can turn parity green by itself.  The remaining tests pin the
machinery so that dropping the files in needs no further code: the C++ writer of the container against the Python reader, and
the consumer run end to end on vectors the oracle wrote into a temporary directory (and made to fail by corrupting them).
Reference functions covered by the key set: src/ORBextractor.cc:410-470, 539-853, 1043-1132; src/Frame.cc:481-655."""
import glob
import importlib
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NO_VECTORS = ("PARITY UNPINNED: no reference vectors in tests/golden/ (ref_*.orbvec).  They come from a real ORB-SLAM2 + OpenCV "
              "build via tools/refvec/dump_reference_vectors.cc - see tests/golden/README.md")


def _refvec():
    import oracle
    oracle.build()
    return importlib.import_module("oracle.refvec")


def _reference_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "ref_*.orbvec")))


def _format(bad):
    return "\n".join("%s: %s  [%s]" % (k, detail, stage) for k, stage, detail in bad[:30])


def test_container_cpp_writer_matches_python_reader(tmp_path):
    rv = _refvec()
    exe, out, pgm = str(tmp_path / "selftest"), str(tmp_path / "self.orbvec"), str(tmp_path / "in.pgm")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "tools", "refvec"),
                           os.path.join(ROOT, "tools", "refvec", "refvec_selftest.cc"), "-o", exe])
    import importlib.util
    spec = importlib.util.spec_from_file_location("write_refvec_inputs", os.path.join(ROOT, "tools", "refvec", "write_refvec_inputs.py"))
    wr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wr)
    img = np.random.default_rng(5).integers(0, 256, (37, 53), dtype=np.uint8)
    img[0, :4] = [10, 32, 9, 13]          # whitespace bytes right behind the header must survive
    wr.write_pgm(pgm, img)
    subprocess.check_call([exe, out, pgm])
    d = rv.read(out)
    assert d["self/image"].tolist() == [[1, 2, 3], [250, 251, 252]] and d["self/image"].dtype == np.uint8
    assert d["self/L3/fast7"].tolist() == [[7, 9, 42], [-1, 1 << 20, 255]]
    assert d["self/meta_f"].tobytes() == np.array([1.2, -0.0, 3.5e-7], "<f4").tobytes()
    assert d["self/crc"].tolist() == [4294967295.0, 0.5]
    assert d["self/empty"].shape == (0, 3)
    assert bytes(d["self/info"]) == b"producer=selftest"
    assert d["self/image_crc"][0] == float(zlib.crc32(bytes([1, 2, 3, 250, 251, 252])))
    assert np.array_equal(d["self/pgm"], img) and d["self/pgm_crc"][0] == float(rv.crc(img))
    # Python writer -> Python reader
    rv.write(str(tmp_path / "py.orbvec"), d)
    d2 = rv.read(str(tmp_path / "py.orbvec"))
    assert list(d2) == list(d) and all(np.array_equal(d[k], d2[k]) and d[k].dtype == d2[k].dtype for k in d)


@pytest.mark.parametrize("case_name", ["tiny_320x240_500", "euroc_752x480_1000"])
def test_consumer_end_to_end_on_oracle_written_vectors(tmp_path, case_name):
    """The consumer on vectors of the right shape (written by the oracle into a temporary directory - NOT reference output and
    never placed in tests/golden/): everything compares equal; one flipped descriptor bit, one moved keypoint, one changed
    pyramid checksum and one changed depth are each reported under their own stage."""
    rv = _refvec()
    case = next(c for c in rv.CASES if c[0] == case_name)
    vec = rv.oracle_vectors(case)
    path = str(tmp_path / ("ref_%s.orbvec" % case_name))
    rv.write(path, vec)
    ref = rv.read(path)
    assert rv.case_of_file(ref) == case and rv.check_inputs(ref, case)
    got = rv.oracle_vectors(case)
    bad, ncmp, missing = rv.compare(ref, got)
    assert not bad and not missing and ncmp >= 8 * 6 + 5, (_format(bad), missing, ncmp)
    p = case_name + "/"
    assert len(rv._kp_fields(ref[p + "keypoints"])) > 300 and ref[p + "L0/fast7"].shape[0] > ref[p + "L0/fast20"].shape[0] > 50
    tampered = []
    g = dict(got); a = g[p + "descriptors"].copy(); a[3, 7] ^= 0x10; g[p + "descriptors"] = a; tampered.append((g, "descriptors"))
    g = dict(got); a = g[p + "L2/keypoints"].copy(); a[0, 0] += 1; g[p + "L2/keypoints"] = a; tampered.append((g, "L2/keypoints"))
    g = dict(got); a = g[p + "L5/crc"].copy(); a[2] += 1; g[p + "L5/crc"] = a; tampered.append((g, "L5/crc"))
    g = dict(got); a = g[p + "L1/octree_direct"][:-1]; g[p + "L1/octree_direct"] = a; tampered.append((g, "L1/octree_direct"))
    if case[4]:
        g = dict(got); a = g[p + "mvDepth"].copy(); a[int(np.argmax(a > 0))] *= np.float32(1.0000001); g[p + "mvDepth"] = a; tampered.append((g, "mvDepth"))
    for g, key in tampered:
        bad, _, _ = rv.compare(ref, g)
        assert [b[0] for b in bad] == [p + key], (key, _format(bad))


def test_consumer_names_the_flavour_a_file_follows(tmp_path):
    """A vector file written under one flavour of the Gaussian is reported as following THAT flavour (not as a bare mismatch), from
    the per-level blur checksums - the two column roundings of OpenCV <= 3.3, the fixed-point Gaussian of >= 3.4.1 with diffused taps -;
    a file with other taps is named by FITTING them to a level it carries in full; anything else is reported as a further variant."""
    rv = _refvec()
    case = next(c for c in rv.CASES if c[0] == "euroc_752x480_1000")
    diffused = "taps:%d,%d,%d,%d" % rv.diffused_taps()
    vecs = {fl: rv.oracle_vectors(case, gauss=fl) for fl in ("half_up", "sse2", diffused)}
    differ = [k for k in vecs["half_up"] if k.endswith("/crc") and not np.array_equal(vecs["half_up"][k], vecs["sse2"][k])]
    assert differ, "the case must hold at least one rounding tie"
    for fl in vecs:
        path = str(tmp_path / "ref_flavour.orbvec")
        rv.write(path, vecs[fl])
        got, res, verdict = rv.identify_flavour(rv.read(path), lambda g: vecs[g])
        assert got == fl and repr(fl) in verdict, verdict
        for other in vecs:
            if other != fl:
                assert {k.rsplit("/", 1)[1] for k, _, _ in res[other][0]} <= {"crc", "descriptors", "descriptors_right", "blur", "mvuRight", "mvDepth"}
    third = dict(vecs["half_up"])
    a = third[differ[0]].copy(); a[2] += 7; third[differ[0]] = a          # blurred pixels that follow no flavour and no tap set
    got, _, verdict = rv.identify_flavour(third, lambda g: vecs[g])
    assert got is None and "FOURTH variant" in verdict, verdict
    # a build with taps nobody guessed: the consumer fits them to the level the small case carries pixel by pixel
    tiny = next(c for c in rv.CASES if c[0] == "tiny_320x240_500")
    odd = "taps:54,49,35,17"
    path = str(tmp_path / "ref_odd.orbvec")
    rv.write(path, rv.oracle_vectors(tiny, gauss=odd))
    cache = {}
    got, _, verdict = rv.identify_flavour(rv.read(path), lambda g: cache.setdefault(g, rv.oracle_vectors(tiny, gauss=g)))
    assert got == odd and "FITTED taps (54, 49, 35, 17)" in verdict, verdict


def test_reference_vectors_pin_the_oracle():
    files = _reference_files()
    if not files:
        pytest.skip(NO_VECTORS)
    rv = _refvec()
    report, followed = [], set()
    for f in files:
        ref = rv.read(f)
        case = rv.case_of_file(ref)
        info = bytes(ref[case[0] + "/info"]).decode()
        assert "producer=reference" in info, "%s was not written by the reference dump program (%s)" % (f, info)
        assert rv.check_inputs(ref, case), "%s: computed on different pixels than the committed synthetic case" % f
        full = int(ref[case[0] + "/meta"][7])
        flavour, res, verdict = rv.identify_flavour(ref, lambda fl: rv.oracle_vectors(case, force_full=full, gauss=fl))
        print("%s: %s" % (os.path.basename(f), verdict))
        assert not any(missing for _, _, missing in res.values())
        if flavour is None:
            report.append("%s (%s): %s\n%s" % (os.path.basename(f), info, verdict, "\n".join(
                "-- flavour %s:\n%s" % (fl, _format(bad)) for fl, (bad, _, _) in res.items())))
        else:
            followed.add(flavour if "EVERY" not in verdict else None)
    followed.discard(None)
    assert len(followed) <= 1, "reference files follow different flavours: %s" % sorted(followed)
    assert not report, "the CPU oracle disagrees with the reference:\n" + "\n".join(report)


@pytest.mark.gpu
def test_reference_vectors_pin_the_hip_path():
    files = _reference_files()
    if not files:
        pytest.skip(NO_VECTORS)
    rv = _refvec()
    be = importlib.import_module("refvec_backends")
    report = []
    for f in files:
        ref = rv.read(f)
        case = rv.case_of_file(ref)
        assert "producer=reference" in bytes(ref[case[0] + "/info"]).decode()
        flavour, res, verdict = rv.identify_flavour(ref, lambda fl: be.hip_vectors(case, gauss=fl))
        print("%s: HIP path: %s" % (os.path.basename(f), verdict))
        if flavour is None:
            report.append("%s: %s\n%s" % (os.path.basename(f), verdict, "\n".join(
                "-- flavour %s:\n%s" % (fl, _format(bad)) for fl, (bad, _, _) in res.items())))
    assert not report, "the HIP path disagrees with the reference:\n" + "\n".join(report)


@pytest.mark.gpu
@pytest.mark.parametrize("case_name", ["tiny_320x240_500", "kitti_1241x376_1000", "natural_1241x376_1000"])
def test_consumer_hip_backend_on_oracle_written_vectors(tmp_path, case_name):
    """The HIP side of the consumer, exercised the only way possible here: on vectors the oracle wrote."""
    rv = _refvec()
    be = importlib.import_module("refvec_backends")
    case = next(c for c in rv.CASES if c[0] == case_name)
    path = str(tmp_path / "v.orbvec")
    rv.write(path, rv.oracle_vectors(case))
    ref = rv.read(path)
    bad, ncmp, missing = be.compare_hip(ref, be.hip_vectors(case))
    assert not bad, _format(bad)
    assert ncmp >= 8 * 3 + 5
    # a file written under the OTHER flavour is recognised as such by oracle and HIP path alike (the quantised case has ties)
    if case_name == "tiny_320x240_500":
        for fl in ("half_up", "sse2", "taps:56,48,34,18", "taps:54,49,35,17"):
            rv.write(path, rv.oracle_vectors(case, gauss=fl))
            ref = rv.read(path)
            got_fl, _, verdict = rv.identify_flavour(ref, lambda g: be.hip_vectors(case, gauss=g))
            assert got_fl is not None and (got_fl == fl or not fl.startswith("taps")), verdict
    # what the HIP path cannot show (primitives it never materialises) is exactly this:
    assert {m.rsplit("/", 1)[1] for m in missing} <= {"fast20", "fast7", "octree_direct"}, missing
