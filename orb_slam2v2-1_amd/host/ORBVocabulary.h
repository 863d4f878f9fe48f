// ORBVocabulary.h — the part of ORB_SLAM2::ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>,
// include/ORBVocabulary.h:28-33) that the BoW hot path uses, executed on an MI355X through include/orbx.h:
// loadFromTextFile and transform(features, BowVector, FeatureVector, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1147-1214, 1351-1436).  The tree descent of every feature
// runs on the GPU; the two std::maps are filled on the host in feature order, exactly as the reference does,
// because BowVector's double sums depend on that order.
#ifndef ORBX_ORBVOCABULARY_H
#define ORBX_ORBVOCABULARY_H
#include <cmath>
#include <map>
#include <string>
#include <vector>
#include "cv_shim.h"
#include "orbx.h"

#ifdef ORBX_HAVE_DBOW2
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#else
namespace DBoW2 {   // Thirdparty/DBoW2/DBoW2/BowVector.{h,cpp}, FeatureVector.{h,cpp}: the members used here
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };
class BowVector : public std::map<WordId, WordValue> {
public:
    void addWeight(WordId id, WordValue v) {   // BowVector.cpp:34-46
        iterator vit = this->lower_bound(id);
        if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
        else this->insert(vit, value_type(id, v));
    }
    void addIfNotExist(WordId id, WordValue v) {   // :50-58
        iterator vit = this->lower_bound(id);
        if (vit == this->end() || (this->key_comp()(id, vit->first))) this->insert(vit, value_type(id, v));
    }
    void normalize(LNorm norm_type) {   // :62-85
        double norm = 0.0;
        iterator it;
        if (norm_type == L1) { for (it = begin(); it != end(); ++it) norm += std::fabs(it->second); }
        else { for (it = begin(); it != end(); ++it) norm += it->second * it->second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (it = begin(); it != end(); ++it) it->second /= norm;
    }
};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {
public:
    void addFeature(NodeId id, unsigned int i_feature) {   // FeatureVector.cpp:29-43
        iterator vit = this->lower_bound(id);
        if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
        else { vit = this->insert(vit, value_type(id, std::vector<unsigned int>())); vit->second.push_back(i_feature); }
    }
};
}  // namespace DBoW2
#endif

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    ORBVocabulary() : mV(NULL), m_k(0), m_L(0), m_scoring(0), m_weighting(0), m_words(0) {}
    ~ORBVocabulary() { if (mV) orbv_destroy(mV); }
    // TemplatedVocabulary::loadFromTextFile (:1351-1436); device: ORBX_DEVICE or 0
    bool loadFromTextFile(const std::string &filename);
    bool empty() const { return m_words == 0; }
    unsigned int size() const { return (unsigned int)m_words; }
    // transform(features, v, fv, levelsup) (:1147-1214)
    void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const;
    orbv_vocabulary_t *handle() const { return mV; }
private:
    ORBVocabulary(const ORBVocabulary &);
    ORBVocabulary &operator=(const ORBVocabulary &);
    orbv_vocabulary_t *mV;
    int m_k, m_L, m_scoring, m_weighting, m_words;
};

}  // namespace ORB_SLAM2
#endif
