// Device context: owns the HIP stream and the HBM buffers of one batch, runs K1 -> K2 -> K3.
// One context per GPU, not shared between host threads.
#pragma once
#include <string>
#include <vector>

#include "batch.hpp"
#include "kernels.hpp"

namespace mp {

struct HostResults {         // device results copied back for the consumer
    // Group / record slots are global indices (allocator << log2 size) + offset (kernels.hpp NPART); only the used prefix
    // of every allocator's sub-range is copied, back to back, and the accessors translate.
    PodVec<WinDyn> win_dyn;      // (not zero-filled before the copies from the device overwrite them)
    PodVec<Group> groups;
    PodVec<GroupSum> gsum;
    PodVec<uint8_t> recs;   // records of rec_stride bytes (HapRecHdr + seq + germ)
    uint32_t seq_cap = 48, rec_stride = 128;
    uint32_t group_part_log2 = 0, rec_part_log2 = 0;
    uint64_t group_prefix[NPART + 1] = {0}, rec_prefix[NPART + 1] = {0};
    size_t gidx(uint64_t slot) const { return size_t(group_prefix[slot >> group_part_log2] + (slot & ((1ull << group_part_log2) - 1))); }
    size_t ridx(uint64_t i) const { return size_t(rec_prefix[i >> rec_part_log2] + (i & ((1ull << rec_part_log2) - 1))); }
    const Group& grp(uint64_t slot) const { return groups[gidx(slot)]; }
    const GroupSum& gsm(uint64_t slot) const { return gsum[gidx(slot)]; }
    const HapRecHdr* rec(uint64_t i) const { return reinterpret_cast<const HapRecHdr*>(recs.data() + ridx(i) * rec_stride); }
    const uint8_t* rec_seq(uint64_t i) const { return recs.data() + ridx(i) * rec_stride + 32; }
    const uint8_t* rec_germ(uint64_t i) const { return recs.data() + ridx(i) * rec_stride + 32 + seq_cap; }
    uint64_t n_group_slots = 0, n_recs = 0;
};

struct RunTiming {
    float k1_ms = 0, k2_ms = 0, k3_ms = 0, k3b_ms = 0, total_ms = 0;
    float k2seq_ms = 0, k2a_ms = 0, k2l_ms = 0, k2w_ms = 0;   // the launches inside k2_ms: sequential replay, admission, lane-per-window, wave-per-window
    float k2win_ms = 0;                                        // the lane- and wave-per-window launches run side by side: their joint wall time
    uint64_t n_group_slots = 0, n_recs = 0, n_groups = 0, n_k3 = 0, n_k3a = 0, n_k3c = 0, n_k3d = 0, n_rec_slots = 0;   // n_k3: groups K3 looked at (the rest were settled by K2l)
    int rows_per_lane = 1;
    uint32_t attempts = 0;
};

class DeviceContext {
  public:
    explicit DeviceContext(int device);
    ~DeviceContext();
    DeviceContext(const DeviceContext&) = delete;
    DeviceContext& operator=(const DeviceContext&) = delete;

    void upload(const Batch& b);   // H2D of the packed batch + plan; allocates outputs
    // One pass of the hot path over the resident batch (inputs in HBM -> results in HBM).
    // Retries with more rows per lane / larger group buffers if the kernels report overflow.
    void run(RunTiming& t);
    void download(HostResults& r); // D2H of the results of the last run()
    void free_batch();
    int device() const { return device_; }
    uint64_t hbm_bytes() const { return hbm_bytes_; }

  private:
    // Host <-> device transfers go through a small ring of PINNED staging buffers (f1; the reference side is the per-gene
    // fetch of src/microphasing.rs:905-942): the DMA engine moves one slot while the host threads fill (H2D) or drain (D2H) the
    // others, so the PCIe link runs at its pinned rate and the page faults of freshly allocated host arrays are taken by several
    // threads beside it - a pageable hipMemcpy stages through the runtime's own buffer on ONE thread. The big host arrays themselves
    // stay pageable: pinning gigabytes costs more than one pass through them.
    struct XferSeg { char* host; char* dev; size_t bytes; };
    static constexpr size_t XFER_SLOTS = 4, XFER_SLOT_BYTES = size_t(32) << 20;
    void* pin_[XFER_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pin_ev_[XFER_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t xfer_stream_ = nullptr;
    double alloc_ms_ = 0; unsigned alloc_calls_ = 0, alloc_reused_ = 0;   // MP_DEBUG: time spent in hipMalloc; blocks taken from the pool
    // Device memory is kept by the context across batches: making ~14 GB of FRESH device memory usable costs ~170 ms at config C (the
    // allocations, then the first copies and kernels that touch it), a fifth of batch_create, for every batch - a service, the gene
    // chunks of config E or of phase_chunked, the shards of a multi-GPU run phase batch after batch on one context. A block a batch
    // gives back stays in the pool; the next batch takes the smallest free block that holds its request and clears it (a fresh
    // allocation reads as zeros; a few ms for a whole batch at HBM speed). Blocks the next batch did not take are freed after its upload,
    // so the pool never holds more than one batch's working set beside the resident one. MP_NO_POOL=1: plain hipMalloc / hipFree.
    struct PoolBlock { void* p; size_t cap; bool in_use; bool used_now; };
    std::vector<PoolBlock> pool_;
    void dfree(void* p);
    void pool_trim(bool all);
    std::vector<XferSeg> pending_up_;     // upload(): the arrays to copy once everything is allocated
    PodVec<ExonW> achunk_exons_;          // upload(): staging of DeviceBatch::achunk_exons
    void xfer(const std::vector<XferSeg>& segs, bool to_device);
    void* dalloc(size_t bytes);
    template <class V> typename V::value_type* up(const V& v);
    void upload_impl(const Batch& b);
    void alloc_outputs();
    void free_outputs();
    int device_ = 0;
    hipStream_t stream_ = nullptr;
    hipStream_t side_[4] = {nullptr, nullptr, nullptr, nullptr};   // the independent launches of the window phase run side by side (run())
    hipEvent_t fork_[2] = {nullptr, nullptr}, join_[4] = {nullptr, nullptr, nullptr, nullptr}, cleared_ = nullptr;
    hipEvent_t k3_fork_ = nullptr, k3_join_ = nullptr, k3_join2_ = nullptr, k3_join3_ = nullptr;    // K3's lists B and C run beside list A on side_[0] / side_[1]
    hipEvent_t ev_[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<void*> allocs_, out_allocs_;
    DeviceBatch d_{};
    uint64_t hbm_bytes_ = 0, out_bytes_ = 0;   // device memory held by the batch; the result buffers' share (re-sized on overflow)
    uint64_t group_cap_ = 0, rec_cap_ = 0;          // NPART << log2
    uint32_t glog_ = 12, rlog_ = 12;                // log2 of one allocator's sub-range
    uint64_t used_g_[NPART] = {0}, used_r_[NPART] = {0};   // slots used by each allocator in the last run()
    int rpl_ = 1;
    uint32_t max_rows_bound_ = 0;
    uint64_t last_slots_ = 0, last_recs_ = 0, last_want_ = 0, last_k3a_ = 0, last_k3b_ = 0, last_k3c_ = 0, last_k3d_ = 0;
};

}  // namespace mp
