/*
 * cvgs_hip.h -- C-ABI of the MI355X-native fused image-preprocessing engine.
 *
 * This is the drop-in boundary for the hot path
 *     crop -> resize(bilinear) -> convertTo/normalize -> cvtColor -> split -> (Circular)Tensor
 * of Libraries-Openly-Fused/cvGPUSpeedup.  The reference has no FFI layer of its own: its
 * host/device boundary is the single call
 *     fk::executeOperations<TF>(cu_stream, iops...)          (reference include/cvGPUSpeedup.cuh:467)
 * into the (un-vendored) FusedKernelLibrary, with every operation parameter passed by value as a
 * kernel argument.  The entry points below are what a binding for that call would bind: the C++
 * facade (cvgpuspeedup_amd/include/cvGPUSpeedup.h, same cvGS:: names as the reference) pattern-
 * matches the compile-time operation list and lowers it to ONE cvgs_chain_desc, and
 * cvgs_execute() launches ONE hand-written HIP kernel (gfx950) for it.
 *
 * Plain C: pointers, sizes and POD structs only.  Every call is asynchronous on the given HIP
 * stream, never synchronises, and is thread-safe (CircularTensor handles excepted: they carry a
 * ring index, as in the reference, include/cvGPUSpeedup.cuh:600-627).  Device memory is only
 * allocated by cvgs_circular_create and cvgs_comm_*, plus one case inside cvgs_execute(_many): a batch
 * with more host descriptors than fit the kernel-argument block (64 planes in a 4 KB block; up to
 * CVGS_KERNARG_PLANES_MAX = 320 planes in a 16 KB block for the batched resize -> planar tensor chain,
 * the reference's benchmark sweep; 52 for warps, 16 destination planes) is written into a slot of a
 * library-owned pool of pinned host buffers (grown on first use, recycled by HIP event, never freed
 * per call) that the kernel reads in place; that case is refused during stream capture -- pass a
 * resident table (cvgs_plane_table_build).  A call whose descriptors travel in the
 * kernel arguments allocates nothing, on the host or on the device, and can be captured.
 *
 * Return value: 0 (CVGS_OK) or a negative cvgs_status; cvgs_last_error() gives a thread-local
 * human-readable message.
 */
#ifndef CVGS_HIP_H
#define CVGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVGS_ABI_VERSION 6
#define CVGS_MAX_OPS 12        /* pointwise stages between the read and the write            */
#define CVGS_MAX_CHANNELS 4
#define CVGS_KERNARG_PLANES 64 /* planes whose descriptors travel inside the kernel arguments */
#define CVGS_KERNARG_PLANES_MAX 320 /* ... and for the batched resize -> planar tensor chain (K1), in a 16 KB argument block */
#define CVGS_MAX_MIRRORS 7     /* extra tensors one chain can write (the 7 peers of an 8-GPU node)  */
#define CVGS_MAX_CHAINS 128    /* chains one cvgs_execute_many launch can fuse                     */
/* Size limits (CVGS_ERR_UNSUPPORTED beyond them; the kernels index inside a row with 32-bit arithmetic): source planes and
 * output planes are at most 2^24 pixels wide and tall; the chroma plane of a 4:2:0 surface starts less than 2 GiB after its
 * luma plane.  Row pitches, plane strides and tensor sizes are 64-bit: a 288 GB tensor is addressable.              */
#define CVGS_MAX_DIM (1 << 24)

typedef void* cvgs_stream_t; /* hipStream_t (0 = the null stream) */

typedef enum cvgs_status {
    CVGS_OK = 0,
    CVGS_ERR_INVALID = -1,     /* malformed descriptor (the reference static_asserts / asserts) */
    CVGS_ERR_UNSUPPORTED = -2, /* well-formed but not implemented on this build                 */
    CVGS_ERR_HIP = -3,         /* a HIP runtime call failed (reference: gpuErrchk)              */
    CVGS_ERR_NO_DEVICE = -4,
    CVGS_ERR_RCCL = -5
} cvgs_status;

/* Element types use OpenCV's numeric encoding so that the cv2cuda shims (reference
 * include/cv2cuda_types.cuh:34-61) need no translation table:
 *   type = depth + ((channels-1) << 3),  depth: 8U=0 8S=1 16U=2 16S=3 32S=4 32F=5 64F=6 16F=7
 * CV_16F (IEEE binary16) is this engine's half-precision hand-off option (the reference has no
 * half type): a storage format only -- per-pixel read source, CAST target (round-to-nearest-even, overflow to
 * +-inf, like cv::saturate_cast<cv::float16_t>) and write type; arithmetic stages need a CAST to CV_32F (or CV_64F) first;
 * a CV_64F value becomes CV_16F through float (two roundings, like the double -> float -> half conversion it spells).   */
#define CVGS_DEPTH_8U 0
#define CVGS_DEPTH_8S 1
#define CVGS_DEPTH_16U 2
#define CVGS_DEPTH_16S 3
#define CVGS_DEPTH_32S 4
#define CVGS_DEPTH_32F 5
#define CVGS_DEPTH_64F 6
#define CVGS_DEPTH_16F 7
#define CVGS_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CVGS_TYPE_DEPTH(t) ((t) & 7)
#define CVGS_TYPE_CN(t) ((((t) >> 3) & 63) + 1)

/* A pitched 2D image view: replaces fk::RawPtr<fk::_2D,T> / fk::Ptr2D<T> as produced by
 * gpuMat2RawPtr2D / gpuMat2Ptr2D (reference include/cvGPUSpeedup.cuh:34-44).  A crop is a view:
 * data = frame + y*step + x*elemSize, same step (reference GpuMat::operator()(Rect),
 * tests/batchresize/test_batchresize_x_split3D.cu:284; cvGS::crop, include/cvGPUSpeedup.cuh:247). */
typedef struct cvgs_image2d {
    const void* data; /* device pointer to pixel (0,0)            */
    int32_t width;    /* pixels                                   */
    int32_t height;   /* rows (NV12: luma rows; UV plane follows) */
    int32_t step;     /* bytes between rows                       */
    /* NV12 kinds only: bytes from `data` to the interleaved UV row that belongs to luma row 0 of this view.
     * 0 = height * step (a whole surface: the UV plane directly below the luma plane).  A CROP of a surface at an
     * even (x, y) is then a view like any other: data = Y + y*step + x, uv_offset = (UV + (y/2)*step + x) - data. */
    int32_t uv_offset;
} cvgs_image2d;

/* ---- read stage (first IOp of the chain) ------------------------------------------------- */
typedef enum cvgs_read_kind {
    /* fk::PerThreadRead<_2D,T>, batched by fk::BatchRead<N,...> (cvGPUSpeedup.cuh:475-583)   */
    CVGS_READ_PIXEL = 0,
    /* fk::Resize<INTER_LINEAR,AR,fk::Read<PerThreadRead<_2D,T>>> wrapped in
     * fk::BatchRead<N,CONDITIONAL_WITH_DEFAULT> (cvGPUSpeedup.cuh:204-245)                    */
    CVGS_READ_RESIZE_LINEAR = 1,
    /* fk::ReadYUV<NV12> + fk::ConvertYUVToRGB<NV12,range,primaries,alpha,floatN>
     * (reference tests/resize/test_fused_resize.cu:50-51)                                     */
    CVGS_READ_NV12 = 2,
    /* the same pair fused as the BackIOp of fk::Resize<INTER_LINEAR>
     * (reference tests/resize/test_fused_resize.cu:141-143)                                   */
    CVGS_READ_NV12_RESIZE_LINEAR = 3,
    /* fk::Warping<fk::WarpType::Affine | Perspective, fk::Read<PerThreadRead<_2D,T>>>, batched like the resize
     * (cvGS::warp, cvGPUSpeedup.cuh:288-442; tests/warping/test_warping_opencv.cu).  For output pixel (x,y) the
     * source position is M*(x,y,1) (perspective: divided by its third component) with M = read.warp_matrices[z],
     * the INVERSE (destination -> source) transform narrowed to float exactly as fk::WarpingParameters holds it
     * (cvGPUSpeedup.cuh:269-284); inside the source [0,w) x [0,h) the value is the INTER_LINEAR interpolation of
     * the resize kinds, outside it is 0; the output type is CV_32F of the source's channels.  Every plane warps into
     * dst_width x dst_height, unless read.warp_dst_sizes gives each plane its own size (the reference's
     * std::array<cv::Size, BATCH> overloads, :381-401): then the write stage must hold one destination image per plane
     * (CVGS_WRITE_PIXEL_2D_BATCH / CVGS_WRITE_SPLIT_2D) of exactly that plane's size; dense tensors need equal sizes.  */
    CVGS_READ_WARP_AFFINE = 4,
    CVGS_READ_WARP_PERSPECTIVE = 5
} cvgs_read_kind;

/* same numeric values as cvGS::AspectRatio (reference include/cvGPUSpeedup.cuh:32) */
typedef enum cvgs_aspect_ratio {
    CVGS_PRESERVE_AR = 0,
    CVGS_IGNORE_AR = 1,
    CVGS_PRESERVE_AR_RN_EVEN = 2,
    CVGS_PRESERVE_AR_LEFT = 3
} cvgs_aspect_ratio;

typedef enum cvgs_yuv_range { CVGS_YUV_FULL = 0, CVGS_YUV_LIMITED = 1 } cvgs_yuv_range;
typedef enum cvgs_yuv_primaries { CVGS_BT601 = 0, CVGS_BT709 = 1, CVGS_BT2020 = 2 /* non-constant luminance */ } cvgs_yuv_primaries;
/* 4:2:0 layouts of the NV12 read kinds (the reference spells the reader as a template on the pixel format,
 * fk::ReadYUV<fk::NV12>, tests/resize/test_fused_resize.cu:50; NV12 is the only format its tests instantiate):
 *   NV12: interleaved chroma plane, U first;  NV21: the same, V first;
 *   I420: planar chroma, a (W/2) x (H/2) U plane with rows of step/2 bytes directly followed by the V plane; YV12: V plane first.
 * The chroma of luma row 0 starts uv_offset bytes after `data` (0 = height * step, the whole surface).  Crops
 * (uv_offset != 0) exist for the interleaved layouts only: a crop of a planar-chroma surface cannot say where its second
 * chroma plane starts (CVGS_ERR_UNSUPPORTED).
 *   P010: the 10-bit decoder surface -- NV12's geometry with 16-bit little-endian samples whose 10 significant bits are the
 *         MOST significant ones (code = sample >> 6); src_type is CV_16UC1, width / height in samples, step and uv_offset in
 *         BYTES.  The conversion works on the 10-bit codes (chroma centre 512, limited range 64..940 / 64..960) and delivers
 *         R, G, B on the 10-bit scale, 0..1023 (alpha = 1023): convertTo CV_16U for a 10-bit image, or scale by 1/1023 in
 *         the chain for a network input.                                                                             */
typedef enum cvgs_yuv_layout { CVGS_YUV_NV12 = 0, CVGS_YUV_NV21 = 1, CVGS_YUV_I420 = 2, CVGS_YUV_YV12 = 3, CVGS_YUV_P010 = 4 } cvgs_yuv_layout;

#define CVGS_READ_FLAG_TABLE_ON_DEVICE 1u /* `src` is a device table made by cvgs_plane_table_build */
/* Device tables in cvgs_execute_many (ABI 6): the caller VOUCHES that nothing this table's planes read lies inside the output of another
 * chain of the same call -- for callers that rewrite their tables in place and cannot state table_src_lo / table_src_hi (below).        */
#define CVGS_READ_FLAG_TABLE_SOURCES_VOUCHED 2u

typedef struct cvgs_read_desc {
    int32_t kind;         /* cvgs_read_kind                                                   */
    int32_t src_type;     /* CV type of the source pixels (NV12: CVGS_MAKETYPE(8U,1))         */
    int32_t batch;        /* planes = grid z (std::array<GpuMat,N>::size())                   */
    int32_t used_planes;  /* planes >= used_planes produce `background` (usedPlanes/activeBatch) */
    const void* src;      /* host: cvgs_image2d[batch]; or device table (flag above)          */
    int32_t dst_width;    /* resize target; PIXEL/NV12 reads: ignored (= source size)         */
    int32_t dst_height;
    int32_t aspect_ratio; /* cvgs_aspect_ratio                                                */
    uint32_t flags;
    float background[4];  /* default value, already in the float type of the read's output    */
    int32_t yuv_range;    /* NV12 kinds only                                                  */
    int32_t yuv_primaries;
    int32_t yuv_alpha;    /* 1: 4-channel output with alpha = 255                             */
    int32_t yuv_layout;   /* cvgs_yuv_layout (0 = NV12)                                       */
    const float* warp_matrices; /* WARP kinds: host, batch x 9 floats (3x3 row-major; affine ignores row 2) */
    /* WARP kinds: host, batch x 2 ints (width, height) = the destination size of every plane (also of planes >= used_planes),
     * or NULL = dst_width x dst_height for all of them.                                                              */
    const int32_t* warp_dst_sizes;
    /* Device tables only (ABI 6): the byte range [table_src_lo, table_src_hi) that holds every byte the table's planes read (chroma rows of
     * 4:2:0 surfaces included), as cvgs_plane_table_hull computes it from the host descriptors the table was built from; NULL / NULL = not
     * stated.  The host cannot see into a device table: cvgs_execute_many compares THIS range with the other chains' outputs before it lets
     * the chains run concurrently in one launch.  A wider range (the whole frame) is always safe.  Ignored by every other call.            */
    const void* table_src_lo;
    const void* table_src_hi;
} cvgs_read_desc;

/* ---- pointwise stages (Unary / Binary IOps) ---------------------------------------------- */
typedef enum cvgs_opcode {
    CVGS_OP_NOP = 0,
    /* fk::SaturateCast<I,O> (cvGS::convertTo, cvGPUSpeedup.cuh:74-129). aux = destination depth. */
    CVGS_OP_CAST = 1,
    /* fk::Binary<fk::Mul/Add/Sub/Div<T>> (cvGS::multiply/add/subtract/divide,
     * cvGPUSpeedup.cuh:131-149); operand[c] = static_cast<float>(cv::Scalar[c])
     * (cvGPUSpeedupHelpers.cuh:38-54), operand_d[c] = the cv::Scalar value itself for CV_64F
     * types.  IEEE fp32 (fp64 on CV_64F values), applied in call order, never merged.  On integer-typed values (since ABI 4;
     * an ENGINE EXTENSION: the reference instantiates fk::Mul<uchar3> etc., whose semantics live in the un-vendored FKL and no
     * reference test uses): the scalar truncated and saturated to the pixel's own type, 64-bit integer arithmetic, division
     * truncating toward zero with x / 0 = 0, the result saturated back to the type; interpreted kernels only.  CV_16F values:
     * CVGS_ERR_UNSUPPORTED.                                                                                              */
    CVGS_OP_MUL = 2,
    CVGS_OP_ADD = 3,
    CVGS_OP_SUB = 4,
    CVGS_OP_DIV = 5,
    /* fk::VectorReorder / the channel-permuting fk::ColorConversion codes
     * (cvGS::cvtColor, cvGPUSpeedup.cuh:151-161).  aux = packed source indices,
     * 2 bits per output channel: out[c] = in[(aux >> 2c) & 3]; channel count unchanged.       */
    CVGS_OP_REORDER = 6,
    /* RGB->RGBA style: out[0..2] = in[(aux>>2c)&3], out[3] = operand[0] (type max). 3 -> 4 ch. */
    CVGS_OP_ADD_ALPHA = 7,
    /* RGBA->RGB style: out[c] = in[(aux>>2c)&3] for c < 3. 4 -> 3 channels.                    */
    CVGS_OP_DROP_ALPHA = 8,
    /* *2GRAY: 0.299 R + 0.587 G + 0.114 B, R = in[aux & 3], B = in[(aux>>4)&3]; integer depths
     * round to nearest even.  3|4 -> 1 channel.                                                */
    CVGS_OP_GRAY = 9,
    /* fk::Cast<I,O> (the reference's warp tests end with fk::Cast<float3,uchar3>, tests/warping/
     * test_warping_opencv.cu:63): static_cast per channel -- float -> integer TRUNCATES toward zero (values outside
     * the destination range, undefined in C++, saturate; NaN -> 0).  aux = destination depth.                  */
    CVGS_OP_CAST_TRUNC = 10
} cvgs_opcode;

typedef struct cvgs_op {
    int32_t opcode;
    int32_t aux;
    float operand[4];    /* operand narrowed to float: used while the value is CV_32F (or an integer depth)   */
    double operand_d[4]; /* the same operand in double: used by arithmetic stages on CV_64F values           */
} cvgs_op;

/* ---- write stage (last IOp of the chain) -------------------------------------------------- */
typedef enum cvgs_write_kind {
    /* fk::PerThreadWrite<_2D,T>: packed pixels into ONE pitched image (cvGS::write(GpuMat),
     * cvGPUSpeedup.cuh:449-452; executeOperations(in,out,...) :489-503).                      */
    CVGS_WRITE_PIXEL_2D = 0,
    /* fk::PerThreadWrite<_3D,T>: packed pixels, dense [plane][y][x] (cvGS::write(GpuMat,Size)
     * :454-457, write(fk::Tensor) :459-462).                                                  */
    CVGS_WRITE_PIXEL_3D = 1,
    /* fk::TensorSplit<T>: dense NCHW  out[z][c][y][x]  (cvGS::split(GpuMat,Size) :185-197)     */
    CVGS_WRITE_TENSOR_SPLIT = 2,
    /* fk::TensorTSplit<T>: dense CNHW out[c][z][y][x]  (cvGS::splitT :199-202)                 */
    CVGS_WRITE_TENSOR_T_SPLIT = 3,
    /* fk::SplitWrite<_2D,T>: C independent pitched planes per batch element
     * (cvGS::split(vector<GpuMat>) / (array<vector<GpuMat>,N>) :163-183)                       */
    CVGS_WRITE_SPLIT_2D = 4,
    /* fk::PerThreadWrite<_2D,T> per batch element: array of pitched images                     */
    CVGS_WRITE_PIXEL_2D_BATCH = 5
} cvgs_write_kind;

typedef struct cvgs_write_desc {
    int32_t kind;     /* cvgs_write_kind                                                       */
    int32_t dst_type; /* CV type of the value being written (depth + channels)                 */
    void* data;       /* tensor kinds and PIXEL_2D: device pointer                             */
    int32_t width;    /* plane width  (pixels)                                                 */
    int32_t height;   /* plane height (rows)                                                   */
    int32_t step;     /* PIXEL_2D: row pitch in bytes; tensor kinds: ignored (dense)           */
    int32_t planes;   /* tensor kinds: number of images N in the tensor (>= read.batch)        */
    /* SPLIT_2D: host array cvgs_image2d[batch*channels], index z*channels+c.
     * PIXEL_2D_BATCH: host array cvgs_image2d[batch].                                         */
    const cvgs_image2d* planes2d;
    /* Tensor kinds (TENSOR_SPLIT / TENSOR_T_SPLIT / PIXEL_3D) only: n_mirrors (<= CVGS_MAX_MIRRORS) further device
     * tensors of the same shape that receive the SAME values at the same element offsets as `data`, in the same kernel
     * (host array of device pointers; NULL / 0 = none).  This is the exchange step of the sharded batched-crop path
     * (SURVEY.md 8e option 2, BASELINE cfg #5): rank r's kernel stores its rows of the [N,C,H,W] tensor into its own
     * copy AND into every peer's copy through P2P-mapped pointers (include/cvgs_rccl.h: cvgs_ipc_* /
     * cvgs_peer_enable), so no separate all-gather moves the data a second time.  No reference counterpart.        */
    void* const* mirrors;
    int32_t n_mirrors;
    int32_t reserved;
} cvgs_write_desc;

/* ---- the fused chain = one kernel launch --------------------------------------------------- */
typedef struct cvgs_chain_desc {
    uint32_t struct_size; /* sizeof(cvgs_chain_desc), for ABI checking */
    uint32_t flags;       /* cvgs_chain_flags                          */
    cvgs_read_desc read;
    int32_t n_ops;
    int32_t reserved;
    cvgs_op ops[CVGS_MAX_OPS];
    cvgs_write_desc write;
} cvgs_chain_desc;

typedef enum cvgs_chain_flags {
    CVGS_CHAIN_DEFAULT = 0,
    /* force the interpreted generic kernel even when a specialised kernel matches (testing)   */
    CVGS_CHAIN_FORCE_GENERIC = 1,
    /* ENABLE_THREAD_FUSION=false of the reference (cvGPUSpeedup.cuh:464): results identical,
     * only disables the multi-pixel-per-thread fast paths.  A tuning / testing knob: the C++
     * facade does not forward executeOperations<false> (the hint exists in the reference because
     * its thread-fused path does not cover every type; here it would only pick a slower kernel) */
    CVGS_CHAIN_NO_THREAD_FUSION = 2
    /* every other bit must be zero (CVGS_ERR_INVALID) */
} cvgs_chain_flags;

/* Library / device ------------------------------------------------------------------------- */
int cvgs_abi_version(void);
const char* cvgs_version_string(void);
const char* cvgs_last_error(void);
/* number of visible HIP devices, or a negative status */
int cvgs_device_count(void);

/* Replaces fk::executeOperations<TF>(stream, iops...) (reference include/cvGPUSpeedup.cuh:467,
 * 480,495,513,524,552,566): validates the chain and enqueues exactly one kernel on `stream`.   */
int cvgs_execute(const cvgs_chain_desc* chain, cvgs_stream_t stream);

/* n_chains INDEPENDENT chains -- distinct source frames, crop lists and output tensors -- in as few launches as
 * possible.  A 50-crop chain moves ~9 MB, about 1 us of HBM time behind a ~1.8 us launch/drain floor;
 * a serving loop with several cameras amortises that floor by submitting its frames together.  Chains whose read is
 * a bilinear resize of 8U/16U/16S/32F pixels, or of NV12 / NV21 surfaces or crops of them (host descriptors), into a planar
 * fp32 / fp16 tensor (the K1 / K4 shapes) -- or, since round 6, a per-pixel read of 8U planes of ONE size (host descriptors) through an fp32 program into a
 * dense fp32 tensor / packed pixels (the reference's batched pointwise chains, tests/batchread/test_batchread_x_write3D.cu:92-96) --, and that agree in
 * everything except read.src / batch / used_planes and write.data / planes (same source type, target size,
 * aspect-ratio mode, background, YUV range / primaries / layout, pointwise stages and operands, write kind and type), are
 * fused into ONE launch of the K1 (or K4) kernel: grid z = chain, grid y = crop.  Results are bit-identical to n_chains separate cvgs_execute calls (same
 * kernel code; tests/test_gpu_many.py).  Any other set of chains is executed one by one, in order -- and so is a set whose
 * chains are NOT independent (two chains write overlapping bytes, or a source view of one -- chroma rows of a 4:2:0
 * surface included -- lies inside another's output: fused chains run concurrently; chains whose plane table lives on the device are checked
 * through the source range they state, read.table_src_lo / table_src_hi from cvgs_plane_table_hull, and a device-table chain that states
 * none runs one by one unless it carries CVGS_READ_FLAG_TABLE_SOURCES_VOUCHED -- ABI 6; until ABI 5 such chains were fused unchecked),
 * a set with a batch beyond 65535, and staged host descriptors under stream capture.  Host
 * descriptors of fused 8-bit chains -- pixels with 3 / 4 channels, or crops of NV12 / NV21 surfaces -- with at most 1024 planes in all (a tick of
 * 16 cameras x 50 crops is 800) travel
 * INSIDE the fused launch's kernel arguments (a 16 KB block up to 256 planes, 52 KB beyond): nothing is staged, nothing is recycled, the
 * call allocates nothing and can be captured as it is; the kernel reads them from device memory (the runtime's argument pool) instead of
 * fetching a pinned table over PCIe (eager tick of 16 x 50 crops: 40.9 -> 39.9 us; the host pays 1-5 us more per call for the larger
 * argument block).  Every other fused launch with host
 * descriptors writes them into a pinned buffer of the stream's own ring that the kernel reads in place (no copy; round 5:
 * the fused kernel itself stores the stream's progress into a pinned word when it starts, and a slot is recycled once that word has reached
 * its launch -- no HIP event behind the launch; a host 8 such launches ahead of ONE stream waits for the oldest of them, at most a few
 * milliseconds, then falls back to the event-tracked descriptor scratch; a stream handle the runtime hands out again after
 * hipStreamDestroy continues the old ring: destroy a stream only once its fused launches have finished); pass device plane tables to
 * make such a fused call capturable; at most
 * CVGS_MAX_CHAINS chains per call.  The reference's closest spelling is the batch sweep of
 * tests/batchresize/test_batchresize_x_split3D.cu:384-392 (one launch per BATCH value).                          */
int cvgs_execute_many(const cvgs_chain_desc* chains, int32_t n_chains, cvgs_stream_t stream);

/* Call before destroying a stream that has carried cvgs_execute_many calls with host descriptors: waits for the stream's work
 * (hipStreamSynchronize -- the stream is about to go anyway) and retires the table ring and progress word the library keeps for that stream
 * HANDLE, so that a later stream the runtime creates with the same handle starts from a clean state and the pinned tables are freed
 * (hipStreamDestroy does not wait for pending work, and the ring is keyed by the handle).  A no-op (CVGS_OK) for streams the
 * library holds nothing for; the C++ facade's cv::cuda::Stream calls it from its destructor.  No reference counterpart.               */
int cvgs_stream_release(cvgs_stream_t stream);

/* Validation only (what the reference checks with static_assert / assert / runtime_error).    */
int cvgs_validate(const cvgs_chain_desc* chain);

/* Name of the kernel cvgs_execute would launch for this chain ("k1_u8c3_swap_mul_sub_div", "generic_inline8", ...)
 * written into buf (NUL terminated).  Introspection for tests and profiles.                    */
int cvgs_kernel_name(const cvgs_chain_desc* chain, char* buf, size_t buf_size);

/* Plane tables: for batches larger than CVGS_KERNARG_PLANES, or when the caller keeps the crop
 * list resident in HBM, the per-plane read parameters live in a device buffer instead of the
 * kernel arguments (the reference is limited to ~50 planes by the 4 KB kernel-parameter block,
 * tests/batchresize/test_batchresize_aspectratio_x_split3D.cu:21-23).  `read->src` must be a
 * host cvgs_image2d[batch]; the function writes cvgs_plane_table_bytes(batch) bytes of
 * position-independent table into host_out, which the caller copies to the device and passes
 * back as read.src with CVGS_READ_FLAG_TABLE_ON_DEVICE.                                         */
size_t cvgs_plane_table_bytes(int32_t batch);
/* A table is bound to the 4:2:0 layout it was built (and validated) with: executing it with another read.yuv_layout is
 * refused for the layouts whose per-plane preconditions differ (P010, I420, YV12: CVGS_ERR_UNSUPPORTED with device tables). */
int cvgs_plane_table_build(const cvgs_read_desc* read, void* host_out);
/* The byte range [*lo, *hi) that holds everything the planes of `read` (host cvgs_image2d[batch], as for cvgs_plane_table_build) read:
 * what a chain that passes the built table states in read.table_src_lo / table_src_hi so that cvgs_execute_many can check it (ABI 6).   */
int cvgs_plane_table_hull(const cvgs_read_desc* read, const void** lo, const void** hi);

/* ---- CircularTensor ------------------------------------------------------------------------
 * Replaces fk::CircularTensor<T,COLOR_PLANES,BATCH,ORDER,CP_MODE> as wrapped by
 * cvGS::CircularTensor (reference include/cvGPUSpeedup.cuh:600-627).                            */
typedef struct cvgs_circular_s* cvgs_circular_t;

typedef enum cvgs_circular_order { CVGS_NEWEST_FIRST = 0, CVGS_OLDEST_FIRST = 1 } cvgs_circular_order;
typedef enum cvgs_color_planes_mode { CVGS_PLANES_STANDARD = 0, CVGS_PLANES_TRANSPOSED = 1 } cvgs_color_planes_mode;

/* elem_type: CV type of ONE tensor element (CV_32FC1 for split tensors, CV_32FC4 for packed).
 * Allocates the output tensor and the internal ring on `device_id` (ctor/Alloc, :605-610).     */
int cvgs_circular_create(cvgs_circular_t* out, int32_t width, int32_t height, int32_t elem_type,
                         int32_t color_planes, int32_t batch, int32_t order, int32_t cp_mode,
                         int32_t device_id);
/* Opt-in variant ("mirrored ring"): the tensor lives in a ring of 2*BATCH image slots and every
 * new frame is written to slot p and to slot p+BATCH, so the BATCH most recent frames are ALWAYS contiguous and in
 * order somewhere in the ring.  An update then costs one fused pass over the new frame (two stores per element) and
 * NO shift traffic: ~50 MB instead of ~800 MB per update at cfg #4.  The price is the API break the reference
 * cannot make: cvgs_circular_data() MOVES with every update (call it after each update), and only the Standard
 * plane order (N,C,H,W) exists.  Tensor contents at data() are identical to the default mode's.                */
#define CVGS_CIRCULAR_MIRRORED 1u
/* CVGS_CIRCULAR_CAPTURABLE: cvgs_circular_update may be captured into a HIP graph (the reference's update is an ordinary stream
 * launch, include/cvGPUSpeedup.cuh:612-622; a serving loop that replays graphs needs it inside them).  The update count then
 * lives in device memory; per-pixel pushes (the form the reference tests) stay ONE fused launch that derives its ring slot from
 * that count, pushes with a resize / NV12 / warp read go through a staging image and a device-indexed shift (one extra pass over
 * ONE image); N captured updates replay as the NEXT N updates.  cvgs_circular_updates() and, for mirrored
 * handles, cvgs_circular_data() read the device-side count and therefore synchronise the device.                        */
#define CVGS_CIRCULAR_CAPTURABLE 2u
int cvgs_circular_create_ex(cvgs_circular_t* out, int32_t width, int32_t height, int32_t elem_type,
                            int32_t color_planes, int32_t batch, int32_t order, int32_t cp_mode,
                            int32_t device_id, uint32_t flags);
/* update(stream, [GpuMat,] iops..., write) (:612-622): `chain` carries the read stage (batch 1),
 * the pointwise stages and the write KIND (TENSOR_SPLIT / TENSOR_T_SPLIT / PIXEL_3D); the write
 * target is the handle's own tensor (write.data is ignored).  The new frame is computed ONCE and stored both into
 * its slot of the ordered tensor and into the history ring, and the BATCH-1 older frames move from the ring to their
 * new slots.  Per-pixel u8 pushes (the form the reference tests) do all of it in ONE kernel launch; pushes with a
 * resize / NV12 / warp read use the chain's own kernel plus one plane-copy kernel (mirrored handles: no copy at all).
 * A default handle cannot be captured into a HIP graph (its ring index is host state): CVGS_ERR_UNSUPPORTED on a capturing
 * stream; handles created with CVGS_CIRCULAR_CAPTURABLE can (above).                                                */
int cvgs_circular_update(cvgs_circular_t ct, const cvgs_chain_desc* chain, cvgs_stream_t stream);
/* data() (:624-626): device pointer of the ordered output tensor; stable for the handle's life
 * (mirrored handles: the window of the LAST update -- it moves).                                 */
void* cvgs_circular_data(cvgs_circular_t ct);
size_t cvgs_circular_bytes(cvgs_circular_t ct);
/* number of updates so far (the reference keeps this host-side ring index private)             */
int64_t cvgs_circular_updates(cvgs_circular_t ct);
int cvgs_circular_destroy(cvgs_circular_t ct);

/* ---- streaming copy ----------------------------------------------------------------------------
 * dst[0..bytes) <- src[0..bytes) with the CircularTensor's plane-copy kernel (non-temporal 16-byte accesses), asynchronous
 * on `stream`.  No reference equivalent: it is the on-box "streaming kernel" copy ceiling SURVEY.md 8(d) asks the
 * roofline fractions to be quoted against (bench.py), and a utility for callers that re-pack tensors.  The two
 * ranges must not overlap.                                                                                      */
int cvgs_stream_copy(void* dst, const void* src, size_t bytes, cvgs_stream_t stream);

/* ---- profiling ranges (reference tests/nvtx.h PUSH_RANGE/POP_RANGE) -------------------------- */
void cvgs_range_push(const char* name);
void cvgs_range_pop(void);

#ifdef __cplusplus
}
#endif
#endif /* CVGS_HIP_H */
