"""Synthetic DBoW2 vocabularies (no vocabulary file ships with the reference) and feature sets for the
BoW tests: a k-ary tree whose children are bit-flipped copies of their parent, in file (= node id) order
with parent < child, some leaves above the last level and some stopped words (weight 0)."""
import numpy as np


def make_vocabulary(rng, k=10, L=3, early_leaf=0.05, stopped=0.05):
    parent, is_leaf, desc, weight, level = [0], [0], [np.zeros(32, np.uint8)], [0.0], [0]
    frontier = [0]
    for lv in range(1, L + 1):
        nxt = []
        for p in frontier:
            for _ in range(k):
                nid = len(parent)
                if p == 0:
                    d = rng.integers(0, 256, 32, dtype=np.uint8)
                else:
                    flips = np.zeros(256, np.uint8)
                    flips[rng.choice(256, max(4, 128 >> lv), replace=False)] = 1
                    d = desc[p] ^ np.packbits(flips)
                leaf = lv == L or (lv >= 2 and rng.random() < early_leaf)
                parent.append(p); desc.append(d); level.append(lv)
                is_leaf.append(1 if leaf else 0)
                weight.append(0.0 if (not leaf or rng.random() < stopped) else float(rng.uniform(0.1, 9.0)))
                if not leaf:
                    nxt.append(nid)
        frontier = nxt
    return dict(k=k, L=L, parent=np.array(parent, np.int32), is_leaf=np.array(is_leaf, np.uint8),
                desc=np.stack(desc), weight=np.array(weight, np.float64), level=np.array(level, np.int32))


def write_text(voc, path, scoring=0, weighting=0):
    """TemplatedVocabulary::saveToTextFile format (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1441-1470)."""
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (voc["k"], voc["L"], scoring, weighting))
        for i in range(1, len(voc["parent"])):
            f.write("%d %d %s %r\n" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]),
                                       float(voc["weight"][i])))


def features_near_words(rng, voc, n, noise_bits=6):
    leaves = np.flatnonzero(voc["is_leaf"] == 1)
    pick = rng.choice(leaves, n)
    d = voc["desc"][pick].copy()
    for i in range(n):
        flips = np.zeros(256, np.uint8)
        flips[rng.choice(256, rng.integers(0, noise_bits + 1), replace=False)] = 1
        d[i] ^= np.packbits(flips)
    far = rng.random(n) < 0.1
    d[far] = rng.integers(0, 256, (int(far.sum()), 32), dtype=np.uint8)
    return d


def intersect(fv_q, fv_c):
    """The merge loop of src/ORBmatcher.cc:176-248: common nodes in increasing id -> CSR lists."""
    nqs, qit, ncs, cit = [0], [], [0], []
    for node in sorted(set(fv_q) & set(fv_c)):
        qit += fv_q[node]; cit += fv_c[node]
        nqs.append(len(qit)); ncs.append(len(cit))
    return np.array(nqs, np.int32), np.array(qit, np.int32), np.array(ncs, np.int32), np.array(cit, np.int32)
