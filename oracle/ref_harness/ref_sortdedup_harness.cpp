// oracle/ref_harness/ref_sortdedup_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Drives the reference's OWN sortmardup classes -- BAMRecord (tbb/bam_record.cpp), SinglePair /
// DoublePair (tbb/pair.cpp) and bitmap (tbb/bitmap.cpp), compiled where they lie under
// /root/reference by oracle/Makefile -- over records handed in as plain arrays.
//
// What is NOT the reference's code here, and why:
//   * sortmardup/main.cpp cannot link (it calls bam_write_idx2, bgzf_flush2, hts_close2,
//     merge_index and hts_idx_finish3, defined nowhere in the tree) and needs TBB, which this image
//     lacks; tbb/bam_parser.cpp needs htslib's sam_parse1, i.e. a built htslib (its build generates
//     config.h/version.h).  Both are unbuildable here.  So the control flow of main.cpp:145-357 for
//     one shuffle thread and the 30-line queue logic of BamParser::pop_record are restated below,
//     while every key, comparison and bit comes from the reference's classes.
//   * bam1_t records are filled in by hand from the arrays (the struct and its accessor macros
//     come from the reference's htslib/sam.h header; no htslib function is called or linked --
//     BAMRecord objects are deliberately never destroyed because ~BAMRecord calls bam_destroy1).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <vector>

#include "pair.h"
#include "bitmap.h"

namespace {

BAMRecord* make_record(uint16_t flag, int32_t tid, int64_t pos, const uint32_t* cigar, uint64_t n_cigar,
                       const uint8_t* qual, uint64_t l_qseq, const char* qname, uint64_t l_name) {
    BAMRecord* r = new BAMRecord;                       // ctor zeroes the bam1_t (bam_record.h:22)
    bam1_t* b = r->get_record();
    const uint64_t l_qname_nul = l_name + 1;
    const uint64_t extranul = (4 - (l_qname_nul & 3)) & 3;
    const uint64_t l_qname = l_qname_nul + extranul;
    const uint64_t l_data = l_qname + n_cigar * 4 + ((l_qseq + 1) >> 1) + l_qseq;
    uint8_t* d = (uint8_t*)calloc(l_data ? l_data : 1, 1);
    memcpy(d, qname, l_name);
    memcpy(d + l_qname, cigar, n_cigar * 4);
    memcpy(d + l_qname + n_cigar * 4 + ((l_qseq + 1) >> 1), qual, l_qseq);
    b->data = d;
    b->l_data = (int)l_data;
    b->m_data = (uint32_t)l_data;
    b->core.flag = flag;
    b->core.tid = tid;
    b->core.pos = pos;
    b->core.l_qname = (uint16_t)l_qname;
    b->core.l_extranul = (uint8_t)extranul;
    b->core.n_cigar = (uint32_t)n_cigar;
    b->core.l_qseq = (int32_t)l_qseq;
    // bam_parser.cpp:54-58
    r->set_pairID((flag & (BAM_FUNMAP | BAM_FSECONDARY | BAM_FSUPPLEMENTARY)) == 0 ? 1 : 0);
    return r;
}

}  // namespace

// out_order[k]  = INPUT index of the k-th record of the output
// out_dup[i]    = 1 iff input record i ends up with BAM_FDUP set by the tool
// out_arrival[k]= INPUT index of the k-th record in arrival order (may be NULL)
extern "C" int ref_sortdedup_run(uint64_t n, const uint16_t* flag, const int32_t* tid, const int64_t* pos,
                                 const uint64_t* cigar_off, const uint32_t* cigar,
                                 const uint64_t* qual_off, const uint8_t* qual,
                                 const uint64_t* qname_off, const char* qname, uint32_t n_targets,
                                 const uint64_t* target_len, uint32_t* out_order, uint8_t* out_dup,
                                 uint32_t* out_arrival) {
    // main.cpp:95-108
    BAMRecord::kTable.clear();
    uint64_t accumulate = 0;
    for (uint32_t i = 0; i < n_targets; i++) { BAMRecord::kTable.push_back(accumulate); accumulate += target_len[i]; }
    BAMRecord::kTable.push_back(accumulate);
    const uint64_t reference_length = BAMRecord::kTable.back();
    bitmap double_pair_indicator(4 * reference_length);              // main.cpp:115

    std::list<std::pair<BAMRecord*, uint32_t>> records;              // BamParser::records + input index
    for (uint64_t i = 0; i < n; i++)
        records.emplace_back(make_record(flag[i], tid[i], pos[i], cigar + cigar_off[i], cigar_off[i + 1] - cigar_off[i],
                                         qual + qual_off[i], qual_off[i + 1] - qual_off[i],
                                         qname + qname_off[i], qname_off[i + 1] - qname_off[i]), (uint32_t)i);

    std::vector<SinglePair*> singles;
    std::vector<DoublePair*> doubles;
    std::vector<std::pair<uint64_t, size_t>> bam_rdd;                // (sort_key, arrival slot), bam_partitioner.cpp:62
    std::vector<BAMRecord*> by_arrival;
    std::vector<uint32_t> arrival_input;
    auto add_record = [&](BAMRecord* r, uint32_t input_idx) {
        bam_rdd.emplace_back(r->sort_key(), by_arrival.size());
        by_arrival.push_back(r);
        arrival_input.push_back(input_idx);
    };

    uint64_t pairID = 1;                                             // main.cpp:35, :146
    while (!records.empty()) {                                       // main.cpp:160-192
        // BamParser::pop_record(pairID)                              bam_parser.cpp:76-83
        auto front = records.front();
        records.pop_front();
        BAMRecord* record1 = front.first;
        if (!record1->ignorable()) record1->set_pairID(pairID);
        // BamParser::pop_record(pairID, hint)                        bam_parser.cpp:85-113
        BAMRecord* record2 = nullptr; uint32_t idx2 = 0;
        if (!record1->ignorable()) {
            for (auto it = records.begin(); it != records.end(); ++it) {
                if (strcmp(record1->qname(), it->first->qname()) != 0) break;
                if (!it->first->ignorable()) {
                    record2 = it->first; idx2 = it->second;
                    records.erase(it);
                    record2->set_pairID(pairID);
                    break;
                }
            }
        }
        if (record2 == nullptr) {
            if (record1->ignorable() == false) singles.push_back(new SinglePair(record1));
            add_record(record1, front.second);
        } else {
            DoublePair* pair = new DoublePair(record1, record2);
            doubles.push_back(pair);
            add_record(record1, front.second);
            add_record(record2, idx2);
            if (pair->get_orientation() == Orientation::FF || pair->get_orientation() == Orientation::RF)
                double_pair_indicator.set(pair->get_record2_prime5_pos());
            else
                double_pair_indicator.set(pair->get_record2_prime5_pos() + reference_length);
            if (pair->get_orientation() == Orientation::FF || pair->get_orientation() == Orientation::FR)
                double_pair_indicator.set(pair->get_record1_prime5_pos());
            else
                double_pair_indicator.set(pair->get_record1_prime5_pos() + reference_length);
        }
        pairID++;
    }

    bitmap duplicate_index(pairID + 1);                              // main.cpp:235
    std::sort(doubles.begin(), doubles.end(), [](DoublePair* a, DoublePair* b) {   // main.cpp:253-264
        if (a->compare_pos_orientation(*b) != 0) return a->compare_pos_orientation(*b) == -1;
        if (a->compare_score(*b) != 0) return a->compare_score(*b) == 1;
        return a->compare_tile_X_Y(*b) != 1;
    });
    for (uint64_t i = 0; i < doubles.size();) {                      // main.cpp:272-279
        uint64_t j;
        for (j = i + 1; j < doubles.size() && doubles[i]->compare_pos_orientation(*(doubles[j])) == 0; j++)
            duplicate_index.set(doubles[j]->get_pairID());
        i = j;
    }
    std::sort(singles.begin(), singles.end(), [](SinglePair* a, SinglePair* b) {   // main.cpp:303-314
        if (a->compare_pos_orientation(*b) != 0) return a->compare_pos_orientation(*b) == -1;
        if (a->compare_score(*b) != 0) return a->compare_score(*b) == 1;
        return a->compare_tile_X_Y(*b) != 1;
    });
    for (uint64_t i = 0; i < singles.size();) {                      // main.cpp:322-338
        if (singles[i]->ignorable()) { i++; continue; }
        auto target = singles[i]->get_prime5_pos();
        if (singles[i]->get_orientation() == Orientation::RR) target += reference_length;
        if (double_pair_indicator.get(target)) duplicate_index.set(singles[i]->get_pairID());
        uint64_t j;
        for (j = i + 1; j < singles.size() && singles[i]->compare_pos_orientation(*(singles[j])) == 0; j++)
            duplicate_index.set(singles[j]->get_pairID());
        i = j;
    }
    std::stable_sort(bam_rdd.begin(), bam_rdd.end(),                 // main.cpp:350-356
                     [](std::pair<uint64_t, size_t> a, std::pair<uint64_t, size_t> b) { return a.first < b.first; });
    for (uint64_t k = 0; k < n; k++) {                               // main.cpp:385-388
        BAMRecord* record = by_arrival[bam_rdd[k].second];
        if (duplicate_index.get(record->get_pairID())) record->mardup();
        out_order[k] = arrival_input[bam_rdd[k].second];
    }
    for (uint64_t a = 0; a < n; a++) {
        out_dup[arrival_input[a]] = (by_arrival[a]->get_record()->core.flag & BAM_FDUP) && !(flag[arrival_input[a]] & BAM_FDUP) ? 1 : 0;
        if (out_arrival) out_arrival[a] = arrival_input[a];
    }
    return 0;      // records, pairs and their data blobs are leaked on purpose (see header comment)
}
