// serving_ticks.cpp -- the reference's call shape in a serving loop, six ways (engine extension of round 4; DESIGN.md 4 "Round 4"):
//   A. cvGS::executeOperations(stream, iops...) per frame, a producer on the stream in front of each call -- the reference's loop
//      (include/cvGPUSpeedup.cuh:464-473), one kernel launch per frame;
//   B. the same frames recorded 16 at a time in a cvGS::ChainBatch and executed on a stream ATTACHED to a descriptor queue with deferred
//      waits: one gate kernel per tick behind the producer, the tick's consumer ordered two ticks later (queue.wait(ticket, stream));
//   C. the same ticks, strictly ordered (the stream is held on every tick), alternating over two attached streams;
//   D. loop A UNCHANGED on a stream attached with cvGS::attachQueueTicks(stream, queue, 16): the calls are recorded and go to the server 16
//      at a time; stream.waitForCompletion() submits what is pending and fences;
//   E. B's ticks on a PLAIN stream: ChainBatch::execute = ONE cvgs_execute_many launch per tick, strictly ordered, no queue;
//   F. loop A unchanged with cvGS::recordTicks(stream, 16): as D with no queue -- one multi-chain launch per 16 recorded calls.
// Host wall clock per frame, including the final synchronise; B's and C's tensors are compared with A's bit for bit.
//   make -C examples && GPU_MAX_HW_QUEUES=3 ./examples/bin/serving_ticks
#include <cvGPUSpeedup.cuh>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

static uint64_t g_s = 0xC0FFEEull;
static uint64_t rnd() {
    g_s += 0x9E3779B97F4A7C15ull;
    uint64_t z = g_s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr int N = 50, FRAMES = 20, TICK = 16, TOTAL = 1920;

struct Frame {
    cv::cuda::GpuMat frame, tensor;
    std::array<cv::cuda::GpuMat, N> crops;
};

// (the clock stops when the STREAMS are done -- hipDeviceSynchronize would also wait for the queue's server to retire, idle_us after
//  its last batch)
template <typename F, typename S>
static double timed_us(F&& fn, S&& sync, int reps = 3) {
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        sync();
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best / TOTAL;
}

int main() {
    const cv::Size up(64, 128);
    const cv::Scalar mul(0.3, 0.3, 0.3), sub(1, 4, 3.2), div(3.2, 0.6, 11.8);
    std::vector<std::unique_ptr<Frame>> fr;
    for (int f = 0; f < FRAMES; ++f) {
        fr.emplace_back(new Frame);
        cv::Mat h(2160, 3840, CV_8UC3);
        for (size_t i = 0; i + 8 <= (size_t)h.step * h.rows; i += 8) {
            const uint64_t v = rnd();
            std::memcpy(h.data + i, &v, 8);
        }
        fr[f]->frame = cv::cuda::GpuMat(h);
        fr[f]->tensor = cv::cuda::GpuMat(N, up.width * up.height * 3, CV_32F);
        for (int i = 0; i < N; ++i) {
            const int w = 32 + (int)(rnd() % 481), hh = 64 + (int)(rnd() % 961);
            fr[f]->crops[i] = fr[f]->frame(cv::Rect((int)(rnd() % (3840 - w + 1)), (int)(rnd() % (2160 - hh + 1)), w, hh));
        }
    }
    auto chain = [&](Frame& f, auto&& sink) {
        sink(cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(f.crops, up, N), cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(), cvGS::multiply<CV_32FC3>(mul),
             cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(f.tensor, up));
    };
    void* scratch = nullptr;
    (void)hipMalloc(&scratch, 4096);
    const size_t tbytes = (size_t)N * up.width * up.height * 3 * sizeof(float);
    auto snapshot = [&]() {
        std::vector<std::vector<uint8_t>> out(FRAMES, std::vector<uint8_t>(tbytes));
        for (int f = 0; f < FRAMES; ++f) {
            (void)hipMemcpy(out[f].data(), fr[f]->tensor.data, tbytes, hipMemcpyDeviceToHost);
            (void)hipMemset(fr[f]->tensor.data, 0xff, tbytes);
        }
        return out;
    };

    // ---- A: one launch per frame -------------------------------------------------------------------------------------------------
    cv::cuda::Stream sa;
    const double us_a = timed_us([&]() {
        for (int i = 0; i < TOTAL; ++i) {
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(sa)); // the producer: a kernel on the stream
            chain(*fr[i % FRAMES], [&](const auto&... iops) { cvGS::executeOperations(sa, iops...); });
        }
    }, [&]() { sa.waitForCompletion(); });
    const auto ref = snapshot();

    // ---- B: ticks of 16 on ONE attached stream, waits deferred two ticks ----------------------------------------------------------
    cvGS::Queue queue(0, 128, 2000.0);
    cv::cuda::Stream sb;
    cvGS::attachQueue(sb, queue, /*deferWait=*/true);
    const double us_b = timed_us([&]() {
        std::vector<uint64_t> tickets;
        cvGS::ChainBatch tick;
        for (int i = 0; i < TOTAL; i += TICK) {
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(sb));
            tick.clear();
            for (int k = 0; k < TICK; ++k) chain(*fr[(i + k) % FRAMES], [&](const auto&... iops) { tick.add(iops...); });
            tick.execute(sb);
            uint64_t t = 0;
            if (cvGS::lastTicket(sb, &t)) tickets.push_back(t);
            if (tickets.size() > 2) queue.wait(tickets[tickets.size() - 3], sb); // the consumer of the tick two ticks back goes here
        }
        cvGS::fence(sb);
    }, [&]() { sb.waitForCompletion(); });
    const auto got_b = snapshot();
    cvGS::detachQueue(sb);

    // ---- C: strictly ordered ticks alternating over two attached streams -----------------------------------------------------------
    cv::cuda::Stream sc[2];
    cvGS::attachQueue(sc[0], queue);
    cvGS::attachQueue(sc[1], queue);
    const double us_c = timed_us([&]() {
        cvGS::ChainBatch tick;
        for (int i = 0, n = 0; i < TOTAL; i += TICK, ++n) {
            cv::cuda::Stream& s = sc[n & 1];
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(s));
            tick.clear();
            for (int k = 0; k < TICK; ++k) chain(*fr[(i + k) % FRAMES], [&](const auto&... iops) { tick.add(iops...); });
            tick.execute(s);
        }
    }, [&]() { sc[0].waitForCompletion(); sc[1].waitForCompletion(); });
    const auto got_c = snapshot();
    cvGS::detachQueue(sc[0]);
    cvGS::detachQueue(sc[1]);

    // ---- D: loop A, unchanged, on a stream attached with recorded ticks -----------------------------------------------------------------
    cv::cuda::Stream sd;
    cvGS::attachQueueTicks(sd, queue, TICK);
    const double us_d = timed_us([&]() {
        for (int i = 0; i < TOTAL; ++i) {
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(sd));
            chain(*fr[i % FRAMES], [&](const auto&... iops) { cvGS::executeOperations(sd, iops...); });
        }
    }, [&]() { sd.waitForCompletion(); });
    const auto got_d = snapshot();
    cvGS::detachQueue(sd);

    // ---- E: the same ticks with NO queue: ChainBatch::execute on a plain stream = ONE cvgs_execute_many launch per tick, strictly ordered ---
    cv::cuda::Stream se;
    const double us_e = timed_us([&]() {
        cvGS::ChainBatch tick;
        for (int i = 0; i < TOTAL; i += TICK) {
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(se));
            tick.clear();
            for (int k = 0; k < TICK; ++k) chain(*fr[(i + k) % FRAMES], [&](const auto&... iops) { tick.add(iops...); });
            tick.execute(se);
        }
    }, [&]() { se.waitForCompletion(); });
    const auto got_e = snapshot();

    // ---- F: loop A unchanged on a stream that records ticks with NO queue (one cvgs_execute_many launch per 16 calls) ------------------------
    cv::cuda::Stream sf;
    cvGS::recordTicks(sf, TICK);
    const double us_f = timed_us([&]() {
        for (int i = 0; i < TOTAL; ++i) {
            (void)hipMemsetAsync(scratch, i & 255, 64, cv::cuda::StreamAccessor::getStream(sf));
            chain(*fr[i % FRAMES], [&](const auto&... iops) { cvGS::executeOperations(sf, iops...); });
        }
    }, [&]() { sf.waitForCompletion(); });
    const auto got_f = snapshot();
    cvGS::stopRecording(sf);

    // what the HOST pays per frame in B: recording the frame's chain (IOps -> descriptor) and the queue's submit (geometry in double, slot)
    double us_record = 0;
    {
        cvGS::ChainBatch tick;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < TOTAL; i += TICK) {
            tick.clear();
            for (int k = 0; k < TICK; ++k) chain(*fr[(i + k) % FRAMES], [&](const auto&... iops) { tick.add(iops...); });
        }
        us_record = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / TOTAL;
    }
    bool same_b = true, same_c = true, same_d = true, same_e = true, same_f = true;
    for (int f = 0; f < FRAMES; ++f) {
        same_b = same_b && std::memcmp(ref[f].data(), got_b[f].data(), tbytes) == 0;
        same_c = same_c && std::memcmp(ref[f].data(), got_c[f].data(), tbytes) == 0;
        same_d = same_d && std::memcmp(ref[f].data(), got_d[f].data(), tbytes) == 0;
        same_e = same_e && std::memcmp(ref[f].data(), got_e[f].data(), tbytes) == 0;
        same_f = same_f && std::memcmp(ref[f].data(), got_f[f].data(), tbytes) == 0;
    }
    const double px = (double)N * up.width * up.height;
    std::printf("A  executeOperations(stream, ...) per frame, producer on the stream          : %6.2f us per frame  %6.1f Gpix/s\n", us_a, px / us_a / 1e3);
    std::printf("B  ChainBatch of %d frames on an attached stream, waits deferred two ticks   : %6.2f us per frame  %6.1f Gpix/s  (== A: %s)\n", TICK, us_b, px / us_b / 1e3, same_b ? "bit for bit" : "DIFFERENT");
    std::printf("C  ChainBatch of %d frames, strictly ordered, two attached streams           : %6.2f us per frame  %6.1f Gpix/s  (== A: %s)\n", TICK, us_c, px / us_c / 1e3, same_c ? "bit for bit" : "DIFFERENT");
    std::printf("D  loop A unchanged on a stream attached with attachQueueTicks(.., %d)        : %6.2f us per frame  %6.1f Gpix/s  (== A: %s)\n", TICK, us_d, px / us_d / 1e3, same_d ? "bit for bit" : "DIFFERENT");
    std::printf("E  ChainBatch of %d frames on a PLAIN stream (one cvgs_execute_many launch)   : %6.2f us per frame  %6.1f Gpix/s  (== A: %s)\n", TICK, us_e, px / us_e / 1e3, same_e ? "bit for bit" : "DIFFERENT");
    std::printf("F  loop A unchanged on a stream with recordTicks(.., %d), no queue             : %6.2f us per frame  %6.1f Gpix/s  (== A: %s)\n", TICK, us_f, px / us_f / 1e3, same_f ? "bit for bit" : "DIFFERENT");
    std::printf("   host: recording one frame's chain in the ChainBatch (IOps -> descriptor)   : %6.2f us per frame\n", us_record);
    (void)hipFree(scratch);
    return same_b && same_c && same_d && same_e && same_f ? 0 : 1;
}
