// Several batches in flight on one GPU behind ONE handle and ONE calling thread (serving throughput).
//
// A single amp_model_infer leaves the chip under-used in three places: while the host reads the detection counts (the mask branch's
// sizes) and the results, in the latency-bound selection / NMS / paste kernels between the convolutions, and in the ramp and tail of
// every launch.  Splitting ONE call into micro-batches does not recover that (measured, tools/exp_halfbatch.py: two halves of a batch
// of 8 on two streams, joined per call, 444 against 456 images/s -- the last micro-batch's latency-bound tail stays exposed and the
// half-size grids fill the chip worse); overlapping CONSECUTIVE batches does (+9 %).  An amp_pipeline owns one worker thread per lane;
// a lane is a model on a context of its own (own HIP stream, own workspace, own result buffers; made by the caller like any other
// model, typically with the same weights).  submit() hands the next batch to the next lane and returns; wait() returns that batch's
// amp_dets.  Lanes are used round-robin, results come back in submission order, and each batch is computed by the same kernels in
// the same order as a plain amp_model_infer: bit-identical results.
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

struct amp_pipeline {
    struct Lane {
        amp_model* model = nullptr;
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        enum { IDLE, QUEUED, DONE } state = IDLE;
        bool stop = false;
        long long ticket = -1;
        // the job
        const uint8_t* imgs = nullptr;
        int on_host = 0, B = 0, H = 0, W = 0;
        std::vector<int> out_h, out_w;
        // its result
        int status = AMP_OK;
        std::string error;
        amp_dets dets;
    };
    std::vector<Lane*> lanes;
    long long next_ticket = 0;
};

namespace {

void lane_loop(amp_pipeline::Lane* L) {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->stop || L->state == amp_pipeline::Lane::QUEUED; });
            if (L->stop) return;
        }
        amp_dets d;
        const int st = amp_model_infer(L->model, L->imgs, L->on_host, L->B, L->H, L->W, L->out_h.empty() ? nullptr : L->out_h.data(),
                                       L->out_w.empty() ? nullptr : L->out_w.data(), &d);
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->status = st;
            L->error = st == AMP_OK ? "" : amp_last_error();      // the message lives in this thread: hand it over
            L->dets = d;
            L->state = amp_pipeline::Lane::DONE;
        }
        L->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int amp_pipeline_create(amp_model* const* models, int depth, amp_pipeline** out) {
    AMP_REQUIRE(models && out && depth >= 1 && depth <= 8, "amp_pipeline_create: bad argument (depth %d)", depth);
    for (int i = 0; i < depth; ++i) {
        AMP_REQUIRE(models[i], "amp_pipeline_create: lane %d has no model", i);
        for (int j = 0; j < i; ++j) AMP_REQUIRE(models[j] != models[i], "amp_pipeline_create: lanes %d and %d are the same model", j, i);
    }
    amp_pipeline* p = new amp_pipeline();
    for (int i = 0; i < depth; ++i) {
        auto* L = new amp_pipeline::Lane();
        L->model = models[i];
        L->worker = std::thread(lane_loop, L);
        p->lanes.push_back(L);
    }
    *out = p;
    return AMP_OK;
}

int amp_pipeline_submit(amp_pipeline* p, const uint8_t* imgs, int imgs_on_host, int B, int H, int W, const int* out_h_h, const int* out_w_h,
                        long long* ticket) {
    AMP_REQUIRE(p && imgs && ticket && B >= 1, "amp_pipeline_submit: bad argument");
    amp_pipeline::Lane* L = p->lanes[(size_t)(p->next_ticket % (long long)p->lanes.size())];
    {
        std::lock_guard<std::mutex> lk(L->mu);
        AMP_REQUIRE(L->state == amp_pipeline::Lane::IDLE, "amp_pipeline_submit: ticket %lld of this lane has not been collected (amp_pipeline_wait): at most %zu batches in flight",
                    L->ticket, p->lanes.size());
        L->imgs = imgs; L->on_host = imgs_on_host; L->B = B; L->H = H; L->W = W;
        L->out_h.clear(); L->out_w.clear();
        if (out_h_h && out_w_h) { L->out_h.assign(out_h_h, out_h_h + B); L->out_w.assign(out_w_h, out_w_h + B); }
        L->ticket = p->next_ticket;
        L->state = amp_pipeline::Lane::QUEUED;
    }
    L->cv.notify_all();
    *ticket = p->next_ticket++;
    return AMP_OK;
}

int amp_pipeline_wait(amp_pipeline* p, long long ticket, amp_dets* out) {
    AMP_REQUIRE(p && out && ticket >= 0 && ticket < p->next_ticket, "amp_pipeline_wait: no such ticket");
    amp_pipeline::Lane* L = p->lanes[(size_t)(ticket % (long long)p->lanes.size())];
    std::unique_lock<std::mutex> lk(L->mu);
    AMP_REQUIRE(L->ticket == ticket && L->state != amp_pipeline::Lane::IDLE, "amp_pipeline_wait: ticket %lld was collected already or has been overtaken", ticket);
    L->cv.wait(lk, [&] { return L->state == amp_pipeline::Lane::DONE; });
    L->state = amp_pipeline::Lane::IDLE;            // the lane may take its next batch; `out` stays valid until that batch is submitted
    if (L->status != AMP_OK) { amp::set_error("%s", L->error.c_str()); return L->status; }
    *out = L->dets;
    return AMP_OK;
}

int amp_pipeline_destroy(amp_pipeline* p) {
    if (!p) return AMP_OK;
    for (auto* L : p->lanes) {
        {
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv.wait(lk, [&] { return L->state != amp_pipeline::Lane::QUEUED; });   // a batch in flight finishes first
            L->stop = true;
        }
        L->cv.notify_all();
        if (L->worker.joinable()) L->worker.join();
        delete L;
    }
    delete p;
    return AMP_OK;
}

}  // extern "C"
