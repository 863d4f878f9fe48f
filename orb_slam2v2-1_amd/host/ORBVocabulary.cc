// ORBVocabulary.cc — see ORBVocabulary.h
#include "ORBVocabulary.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ORB_SLAM2 {

bool ORBVocabulary::loadFromTextFile(const std::string &filename) {
    if (mV) { orbv_destroy(mV); mV = NULL; m_words = 0; }
    const int device = std::getenv("ORBX_DEVICE") ? std::atoi(std::getenv("ORBX_DEVICE")) : 0;
    if (orbv_load_text(filename.c_str(), device, &mV) != ORBX_OK) {
        std::fprintf(stderr, "ORBVocabulary::loadFromTextFile: %s\n", orbx_last_error());
        return false;
    }
    int nn = 0;
    orbv_info(mV, &m_k, &m_L, &m_scoring, &m_weighting, &nn, &m_words);
    return true;
}

void ORBVocabulary::transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv,
                              int levelsup) const {
    v.clear();
    fv.clear();
    if (empty()) return;
    const int n = (int)features.size();
    if (n == 0) return;
    std::vector<uint8_t> desc((size_t)32 * n);
    for (int i = 0; i < n; i++) std::memcpy(&desc[(size_t)32 * i], features[i].ptr(0), 32);
    std::vector<int32_t> word(n), node(n);
    std::vector<double> weight(n);
    if (orbv_transform(mV, desc.data(), n, levelsup, word.data(), node.data(), weight.data()) != ORBX_OK) {
        std::fprintf(stderr, "ORBVocabulary::transform: %s\n", orbx_last_error());
        return;
    }
    // mustNormalize: L1 / L2 / chi-square / KL / Bhattacharyya scoring normalise, the dot product does not (ScoringObject.h:73-90)
    const bool must = m_scoring != DBoW2::DOT_PRODUCT;
    const DBoW2::LNorm norm = m_scoring == DBoW2::L2_NORM ? DBoW2::L2 : DBoW2::L1;
    if (m_weighting == DBoW2::TF || m_weighting == DBoW2::TF_IDF) {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) {   // not stopped
                v.addWeight((DBoW2::WordId)word[i], weight[i]);
                fv.addFeature((DBoW2::NodeId)node[i], (unsigned int)i);
            }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {   // IDF || BINARY
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) {
                v.addIfNotExist((DBoW2::WordId)word[i], weight[i]);
                fv.addFeature((DBoW2::NodeId)node[i], (unsigned int)i);
            }
    }
    if (must) v.normalize(norm);
}

}  // namespace ORB_SLAM2
