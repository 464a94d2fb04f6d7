/*
 * cvgs_hip_ext.h -- ENGINE EXTENSIONS of libcvgs_hip.so that have no counterpart in the reference's interface: the device-side
 * descriptor queue (cvgs_queue_*: an opt-in submission path, frozen since round 5) and the device-side arrival flags of the sharded
 * batched-crop path (cvgs_exchange_*: BASELINE cfg #5, SURVEY.md 8e option 2).  The drop-in boundary -- what replaces
 * fk::executeOperations and fk::CircularTensor (reference include/cvGPUSpeedup.cuh:464-627) -- is include/cvgs_hip.h alone; nothing
 * there depends on this file.  Same conventions: plain C, asynchronous on the given stream, 0 or a negative cvgs_status.
 */
#ifndef CVGS_HIP_EXT_H
#define CVGS_HIP_EXT_H

#include "cvgs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-side descriptor queue (ABI 4) ---------------------------------------------------------------------
 * The reference submits one kernel per executeOperations call (include/cvGPUSpeedup.cuh:464-473 -> fk::executeOperations:
 * one TransformDPP launch).  On MI355X a 50-crop batch is ~1 us of HBM time behind a ~1.8 us launch/drain boundary, and the
 * waves of ONE launch load, then store, all at the same time.  A queue keeps the call shape -- one submit per
 * frame, same chain descriptor -- and removes the boundary: a resident server grid takes batches from a ring; its
 * workgroups walk from batch to batch without a grid-wide barrier, so batch k+1's loads overlap batch k's stores.
 *   cvgs_queue_create   one queue per device (`device` < 0: the calling thread's current device); `depth` ring slots (0 = 128, at most 256); `idle_us`: the server retires
 *                       itself after this long without work (0 = 200 us) and the next submit starts a new one, so the grid
 *                       never outlives its work; a batch without progress for 250 ms (environment: CVGS_QUEUE_STALL_MS) is reported
 *                       as CVGS_ERR_HIP, not waited for -- every workgroup of the server must be resident, so other kernels of the
 *                       process must not hold the whole chip for longer than that.
 *   cvgs_queue_submit   asynchronous; the chain must be K1's hot shape (batched 8UC3 / 8UC4 -- or 16UC3 / 16UC4 / 16SC3 / 16SC4 -- bilinear resize -> [RGB<->BGR]
 *                       mul, sub, div [-> convertTo CV_16F] -> fp32 / fp16 NCHW / CNHW tensor, host descriptors; a ring slot holds 74 planes, larger batches take
 *                       consecutive slots behind ONE ticket) or the same behind crops of
 *                       NV12 / NV21 / P010 decoder surfaces (CVGS_READ_NV12_RESIZE_LINEAR, 3 channels; letterboxing and default planes
 *                       included): anything else returns CVGS_ERR_UNSUPPORTED and belongs to cvgs_execute.  A queue serves ONE
 *                       of the four kinds (8-bit pixels, 16-bit pixels, NV12 / NV21 surfaces, P010 surfaces) -- its first submit decides, the others are then CVGS_ERR_UNSUPPORTED (each kind
 *                       has its own server grid; create a second queue).  The sources must be complete when submit is
 *                       called (the server is not ordered behind any stream); results are bit-identical to cvgs_execute.
 *   cvgs_queue_wait     host waits for a ticket AND every batch submitted before it (tickets are handed out in submit order; the
 *                       server completes batches in any order -- their tasks are spread over its workers -- so the wait looks at every
 *                       batch up to the ticket that it has not yet seen complete); cvgs_queue_stream_wait makes a HIP stream
 *                       wait for the same set instead: ONE one-wave polling kernel on that stream (k1q_wait; an unsatisfied
 *                       hipStreamWaitValue64 on device memory costs ~1.6 ms here), the
 *                       consumer's kernels enqueued behind it see the tensors.  RELEASE ON FAILURE: the gate kernel of a stream-ordered
 *                       submit and the wait kernel return -- and so release the stream -- when the queue's error word is set or their
 *                       time limit expires (the 10 s gate limit / the stall limit); the tensor may then be incomplete and NOTHING on the
 *                       stream says so.  The failure is reported by the next cvgs_queue_wait / submit / stats call (error word != 0):
 *                       a consumer that must not run on an incomplete tensor checks cvgs_queue_wait(ticket) == CVGS_OK first.
 * Environment: CVGS_QUEUE_G = worker workgroups (default 2 per CU - 1 for 8-bit pixel crops, 3 per CU - 1 for the other kinds; the flags' bits 16..27 say the same per queue),
 * CVGS_QUEUE_STALL_MS = the stall watchdog, CVGS_QUEUE_DEBUG=1 = a state dump when it fires (=2: gate timestamps for tools/probes).  Frozen since round 5: no new kinds, no new knobs.
 * Submits from several host threads are serialised by a mutex (tickets are handed out in submit order).  cvgs_queue_destroy
 * waits (at most 2 s) for the batches in flight, then retires the server.  Whether the host writes the ring straight into device
 * memory is decided without a fault (large-BAR attribute + /proc/self/maps + a read-back; flags bit 0 asks for the staged ring).
 * RUNTIME NOTE: while a server grid is alive, kernels of every stream whose hardware queue shares the server queue's command-processor
 * pipe dispatch at ~27 us each instead of ~3.4 us (measured: one stream in four on the default runtime's 4 hardware queues, none with
 * GPU_MAX_HW_QUEUES <= 3; tools/probes/server_vs_streams.py).  Processes that keep a queue alive beside other streams should set
 * GPU_MAX_HW_QUEUES=3 before the HIP runtime initialises; the server retires idle_us after its last batch.
 * No reference counterpart.                                                                                           */
typedef struct cvgs_queue_s* cvgs_queue_t;
int cvgs_queue_create(cvgs_queue_t* out, int32_t device, int32_t depth, double idle_us, uint32_t flags);
int cvgs_queue_submit(cvgs_queue_t q, const cvgs_chain_desc* chain, uint64_t* ticket);
/* n submits in one call (a serving loop's burst); *last_ticket = the ticket of chains[n-1] */
int cvgs_queue_submit_many(cvgs_queue_t q, const cvgs_chain_desc* const* chains, int32_t n, uint64_t* last_ticket);
/* Stream-ordered submit (ABI 5) -- the reference's contract on the queue: cvGS::executeOperations(stream, iops...) is "asynchronous on
 * the given stream" (include/cvGPUSpeedup.cuh:464-473; cv::cuda::StreamAccessor::getStream at :466).  The batch is ordered BEHIND
 * everything already enqueued on `stream` (the decoder / kernel that writes the frame need not be synchronised with the host) and,
 * unless CVGS_QUEUE_SUBMIT_DEFER_WAIT, everything enqueued on `stream` AFTER the call is ordered behind the batch's tensor.  Cost: ONE
 * one-wave kernel on `stream` per call (it opens the batch's gate in the ring when the stream gets there, then holds the stream on the
 * batch's completion word); no host synchronisation, no event, no second stream operation.  Workers that draw a task of a batch whose
 * gate is still closed wait there; other batches proceed.  A gate that stays closed for 10 s is reported
 * as an error (word 3), not waited for.
 *   CVGS_QUEUE_SUBMIT_DEFER_WAIT  the stream is NOT held: the caller orders the consumer itself with cvgs_queue_stream_wait(q, ticket,
 *                                 stream) -- several batches of ONE stream can then be in flight at once (a strictly ordered stream has
 *                                 one, because the next gate sits behind the previous wait).  Until that wait the batch's SOURCES are
 *                                 in flight too: work enqueued on the stream behind the call runs concurrently with the batch and must
 *                                 not rewrite them (a strictly ordered call protects both sides).
 *   CVGS_QUEUE_SUBMIT_HYBRID      latency policy ("never slower than without a queue"): stream order costs one launch per gate, so a
 *                                 gate in front of fewer than 8 chains (CVGS_QUEUE_SUBMIT_MIN_GROUP(n) changes the 8) never beats
 *                                 launching them -- such calls, a batch nothing in flight could overlap with, and any chain the
 *                                 server does not take are launched DIRECTLY on `stream` as cvgs_execute would (*ticket =
 *                                 CVGS_QUEUE_TICKET_DIRECT).  Measured: a lone strict stream 14-16 us per batch on the server against
 *                                 8-9 us as launches; ticks of 16 frames (cvgs_queue_submit_many_on) 2.5 us per frame against 8.9.
 * Not capturable (the ring is written at submit time): CVGS_ERR_UNSUPPORTED on a capturing stream (with HYBRID: the direct launch is
 * captured instead).  At most 74 planes per call.                                                                          */
#define CVGS_QUEUE_SUBMIT_DEFER_WAIT 1u
#define CVGS_QUEUE_SUBMIT_HYBRID 2u
#define CVGS_QUEUE_SUBMIT_MIN_GROUP(n) (((uint32_t)(n) & 0xffu) << 8) /* with HYBRID: the smallest group the server takes (0 = 8) */
#define CVGS_QUEUE_TICKET_DIRECT (~(uint64_t)0)
int cvgs_queue_submit_on(cvgs_queue_t q, const cvgs_chain_desc* chain, cvgs_stream_t stream, uint32_t flags, uint64_t* ticket);
/* n chains (<= CVGS_QUEUE_MAX_GROUP) behind ONE gate kernel: the pictures of one tick -- several cameras' frames written by the work in
 * front of the call, several crop lists of one frame -- are ordered behind `stream` together, overlap on the server, and (unless
 * DEFER_WAIT) the stream is held until ALL of them are complete: one launch per tick instead of one per chain.  The stream-ordered
 * counterpart of cvgs_execute_many / cvgs_queue_submit_many.  A wait on *last_ticket covers the group.  The chains of a group run
 * CONCURRENTLY on the server, so they must be independent -- no chain may write what another chain of the group reads or writes (checked
 * for tensor targets against each other and against host-described sources, as cvgs_execute_many does; sources in caller-owned device
 * tables cannot be checked): a dependent group is launched one by one, in order, under HYBRID and is CVGS_ERR_UNSUPPORTED without it.
 * A caller stream created at the HIGHEST stream priority (the server's own) may share the server's hardware queue, where its gate kernel
 * could never start: such streams are not taken by the server (HYBRID: direct launches; otherwise CVGS_ERR_UNSUPPORTED).  With HYBRID, groups below the
 * minimum and chains the server does not take are launched one by one on the stream, in order -- and a STRICTLY ordered group (no
 * DEFER_WAIT, no explicit MIN_GROUP) is ONE multi-chain launch on the stream (cvgs_execute_many; *last_ticket = CVGS_QUEUE_TICKET_DIRECT):
 * the stream is held until the group is complete either way, and measured (ticks of 16 frames, a producer on the stream) the launch
 * serves a frame in 2.4-3.0 us where gate + server take 2.7-3.4, with nothing resident beside the consumer.  The server keeps what it is
 * better at: deferred waits (2.4-2.5 us per frame on ONE stream) and host tickets (cvgs_queue_submit, 2.15).               */
#define CVGS_QUEUE_MAX_GROUP 64
int cvgs_queue_submit_many_on(cvgs_queue_t q, const cvgs_chain_desc* const* chains, int32_t n, cvgs_stream_t stream, uint32_t flags, uint64_t* last_ticket);
/* After a wait / submit has reported CVGS_ERR_HIP because the server's watchdog fired (another kernel held the chip beyond
 * CVGS_QUEUE_STALL_MS, a gate never opened): waits for the failed server to leave, declares the batches that were in flight lost
 * (*lost = how many; waits on their tickets return CVGS_ERR_HIP, streams waiting for them are released, their tensors may be
 * incomplete), resets the protocol state and clears the error -- the next submit starts a fresh server.  A no-op on a healthy queue. */
int cvgs_queue_recover(cvgs_queue_t q, uint64_t* lost);
int cvgs_queue_wait(cvgs_queue_t q, uint64_t ticket, double timeout_s);
int cvgs_queue_stream_wait(cvgs_queue_t q, uint64_t ticket, cvgs_stream_t stream);
/* out[8]: submitted, completed, server launches, feeder rounds and lifetime (100 MHz ticks) of the last retired server,
 * worker workgroups, ring slots, error word */
int cvgs_queue_stats(cvgs_queue_t q, uint64_t* out8);
/* the hipStream_t the server grid is launched on (for HIP events / profilers; do not enqueue work behind a live server) */
cvgs_stream_t cvgs_queue_stream(cvgs_queue_t q);
/* out[16], of the last RETIRED server, 100 MHz ticks / counts: feeder {rounds with copies, slots, 0, 0 (reserved)},
 * monitor {scans, scan ticks, ring waits of the host, batches complete behind an incomplete oldest one (mean)}, worker 0 {tasks, find ticks, rows ticks, drain
 * ticks, idle polls}, host ns per submit spent waiting for a ring slot, stream-ordered submits taken by the server | launched directly << 32, gate trace address (probes) */
int cvgs_queue_profile(cvgs_queue_t q, uint64_t* out16);
int cvgs_queue_destroy(cvgs_queue_t q);

/* ---- device-side arrival flags for the sharded batched-crop path (ABI 4; BASELINE cfg #5, SURVEY.md 8e option 2) ----------
 * With the P2P fused write (cvgs_write_desc.mirrors) every rank's K1 launch stores its rows of the [N,C,H,W] tensor into every
 * peer's copy; what remains of the exchange is knowing when all rows of a step have landed.  These two calls keep that on the
 * device: flags are 8-byte words the ranks place in their IPC-shared allocations (include/cvgs_rccl.h: cvgs_ipc_*), one word per
 * source rank, at least 128 bytes apart.
 *   cvgs_exchange_signal  enqueue behind the step's launch: stores `value` (the step number, monotonic) into the n given words --
 *                         this rank's word in each peer's flag block (pointers valid in THIS process).  The kernel boundary in
 *                         front of it makes the step's rows visible before the flags.
 *   cvgs_exchange_wait    the stream waits until each of the n given words (this rank's own flag block: one word per peer)
 *                         is >= `value`; after `timeout_ms` (0 = 2000) it gives up, stores {1, index of a flag that was behind}
 *                         into err_words[0..1] (device or pinned memory, may be NULL) and lets the stream continue -- a lost
 *                         peer is reported, never waited for; while err_words[0] is non-zero every later wait / step behind the same
 *                         error words returns at once (ONE timeout per lost peer, not one per step; clear the words to re-arm).  n <= 16.
 * `step_counter` (device memory, 8 bytes, zero-initialised by the caller; NULL = use `value`): the step number then lives on the
 * device -- signal advances *step_counter and publishes the new count, wait waits for *step_counter - lag (and for nothing while
 * the count is <= lag) -- so that a whole sequence of steps can be captured into ONE HIP graph and replayed (a captured constant
 * would repeat).  With n == 0 and a counter, cvgs_exchange_signal still advances it; cvgs_exchange_step with n == 0 enqueues NOTHING and
 * leaves the counter alone (one rank has nobody to tell and nothing to wait for -- the two differ on purpose).
 * No collective and no host round trip per step.  No reference counterpart (the reference is single-GPU).                 */
int cvgs_exchange_signal(void* const* peer_flag_words, int32_t n, uint64_t value, uint64_t* step_counter, cvgs_stream_t stream);
int cvgs_exchange_wait(const void* const* own_flag_words, int32_t n, uint64_t value, const uint64_t* step_counter, uint64_t lag, double timeout_ms,
                       void* err_words, cvgs_stream_t stream);
/* signal + lagged wait as ONE launch per step (what a steady loop enqueues behind each K1): advances *step_counter, publishes it
 * into peer_flag_words[0..n), then waits until own_flag_words[0..n) have reached the count - lag.  n == 0 (one rank): nothing is
 * enqueued.  About one kernel boundary (~2 us) per step, against >= 20 us of link time per step on 8 GPUs.                      */
int cvgs_exchange_step(void* const* peer_flag_words, const void* const* own_flag_words, int32_t n, uint64_t* step_counter, uint64_t lag,
                       double timeout_ms, void* err_words, cvgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CVGS_HIP_EXT_H */
