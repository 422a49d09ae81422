// Internal: the opaque C-ABI handle types (include/microphaser_hip.h) and the exception guard, shared by capi*.cpp.
#pragma once
#include <mutex>
#include <memory>
#include <string>

#include "../../include/microphaser_hip.h"
#include "batch.hpp"
#include "consume.hpp"
#include "device.hpp"
#include "filter.hpp"
#include "pep.hpp"
#include "synth.hpp"

using namespace mp;

struct mp_ctx {
    std::unique_ptr<DeviceContext> dev;
    const void* resident = nullptr;   // the batch whose buffers the device context currently holds (one at a time)
    const void* last_run = nullptr;   // the batch the device results belong to
    std::string err;
};
struct mp_dataset {
    Dataset ds;
};
struct mp_batch {
    Batch batch;                   // GeneHost::input points into the data set, which must outlive the batch
    const ReadStore* reads = nullptr;
    bool uploaded = false, ran = false;
    RunTiming timing;
    uint64_t sum_wlen = 0, sum_cols = 0;  // cached for the byte accounting
    uint64_t w_steps = 0, w_wins = 0;     // steps / printing steps replayed window-parallel
    bool w_wins_known = false;
    uint64_t adm_wave = 0, adm_lane = 0;  // (exon, read) entries K2a writes an AdmEntry / a RowRec for
    bool adm_known = false;
};
struct mp_results {
    PhasedStreams out;   // normal mode: fasta / tsv / n_windows are filled, normal_fasta stays empty
};
struct mp_filtered {
    FilterResult res;
};
struct mp_peptides {
    PeptideResult res;
    std::string bin;            // bincode image of the set (src/peptides.rs:183); an EMPTY set still encodes as its 8-byte length
    std::once_flag bin_once;    // built eagerly by mp_build_reference*, else on the first mp_peptides_binary (any thread)
};

namespace {
template <class F>
int guarded(mp_ctx* ctx, F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        if (ctx) ctx->err = e.what();
        return 1;
    } catch (...) {
        if (ctx) ctx->err = "unknown error";
        return 1;
    }
}
DeviceContext& need_device(mp_ctx* ctx) {
    if (!ctx->dev) throw Error("this context has no GPU (created with device -1): the phasing kernels need an MI355X, there is no CPU fallback");
    return *ctx->dev;
}
}  // namespace
