"""Randomised parity stress of the matchers: python tools/stress_matchers.py [rounds]
Runs the scenarios of tests/test_match_gpu.py again and again with shifted random seeds (every
np.random.default_rng(seed) inside a scenario becomes default_rng(seed + 1000 * round)), on both the
speculative and the exact implementation of the guided searches.  Not part of the pytest suite."""
import sys, os, importlib, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("orb_slam2v2-1_amd")
synth = importlib.import_module("orb_slam2v2-1_amd.synth")
import oracle
oracle.build()
import test_match_gpu as T

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
orig_rng = np.random.default_rng
bad = 0
t0 = time.time()
for rd in range(1, rounds + 1):
    np.random.default_rng = lambda s=None, _o=1000 * rd: orig_rng(None if s is None else s + _o)
    import bow_scene as bs
    rng = np.random.default_rng(41)
    voc = bs.make_vocabulary(rng, k=10, L=3)
    bow = (bs, rng, voc, oracle.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]),
           pkg.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]))
    cases = []
    for path in ("fast", "exact"):
        for f in (T.test_search_for_initialization, T.test_search_by_projection_mappoints, T.test_search_by_projection_frame,
                  T.test_guided_search_heavy_contention, T.test_search_by_projection_keyframe, T.test_match_windows_generic_paths_agree,
                  T.test_search_local_points_fused):
            cases.append((f.__name__ + "[" + path + "]", path, lambda f=f, path=path: f(pkg, oracle, synth, path)))
        for dist in (False, True):
            cases.append(("sim3[%s,%s]" % (path, dist), path, lambda path=path, dist=dist: T.test_search_by_projection_sim3(pkg, oracle, synth, path, dist)))
    for dist in (False, True):
        for gate in (False, True):
            cases.append(("best_in_windows[%s,%s]" % (dist, gate), "fast", lambda dist=dist, gate=gate: T.test_best_in_windows(pkg, oracle, synth, dist, gate)))
    cases.append(("distinctive", "fast", lambda: T.test_distinctive_descriptors(pkg, oracle)))
    cases.append(("frustum", "fast", lambda: T.test_is_in_frustum(pkg, oracle, synth)))
    for v in ("kf_frame", "kf_kf"):
        cases.append(("bow[%s]" % v, "fast", lambda v=v: T.test_search_by_bow(pkg, oracle, bow, v)))
    cases.append(("triangulation", "fast", lambda: T.test_search_for_triangulation(pkg, oracle, bow)))
    for name, path, fn in cases:
        pkg.lib().orbm_set_thread_option(2, 1 if path == "exact" else 0)
        try:
            fn()
        except AssertionError as e:
            msg = str(e).splitlines()[0] if str(e) else "assert"
            # scenario-statistics asserts (e.g. "on > 200") are not parity failures
            soft = ("> " in msg and "==" not in msg) and "Arrays are not equal" not in msg
            print(("note" if soft else "MISMATCH"), "round", rd, name, msg[:100])
            if not soft:
                bad += 1
        except Exception:
            bad += 1
            print("ERROR round", rd, name)
            traceback.print_exc(limit=2)
        finally:
            pkg.lib().orbm_set_thread_option(2, 0)
np.random.default_rng = orig_rng
print("matcher stress: %d rounds, %d failures, %.1f s" % (rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
