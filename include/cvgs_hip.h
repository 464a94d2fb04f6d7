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

#define CVGS_ABI_VERSION 5
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
 * fp32 / fp16 tensor (the K1 / K4 shapes), and that agree in
 * everything except read.src / batch / used_planes and write.data / planes (same source type, target size,
 * aspect-ratio mode, background, YUV range / primaries / layout, pointwise stages and operands, write kind and type), are
 * fused into ONE launch of the K1 (or K4) kernel: grid z = chain, grid y = crop.  Results are bit-identical to n_chains separate cvgs_execute calls (same
 * kernel code; tests/test_gpu_many.py).  Any other set of chains is executed one by one, in order -- and so is a set whose
 * chains are NOT independent (two chains write overlapping bytes, or a host-described source view of one -- chroma rows of a 4:2:0
 * surface included; chains whose plane table lives on the device are NOT checked: the caller vouches for them -- lies inside another's
 * output: fused chains run concurrently), a set with a batch beyond 65535, and staged host descriptors under stream capture.  Host
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
 *                       hipStreamWaitValue64 on device memory costs ~1.6 ms here -- CVGS_QUEUE_WAITVALUE=1 keeps that spelling), the
 *                       consumer's kernels enqueued behind it see the tensors.  RELEASE ON FAILURE: the gate kernel of a stream-ordered
 *                       submit and the wait kernel return -- and so release the stream -- when the queue's error word is set or their
 *                       time limit expires (the 10 s gate limit / the stall limit); the tensor may then be incomplete and NOTHING on the
 *                       stream says so.  The failure is reported by the next cvgs_queue_wait / submit / stats call (error word != 0):
 *                       a consumer that must not run on an incomplete tensor checks cvgs_queue_wait(ticket) == CVGS_OK first.
 * Tuning hooks (environment): CVGS_QUEUE_G = worker workgroups (default 2 per CU - 1 for 8-bit pixel crops, 3 per CU - 1 for the other kinds; the flags' bits 16..27 say the same per queue),
 * CVGS_QUEUE_DEEP_ROWS = rows per task of a deep queue (default 64 / 128 by depth), CVGS_QUEUE_STALL_MS, CVGS_QUEUE_STAGED=1, CVGS_QUEUE_DEBUG=1.
 * Submits from several host threads are serialised by a mutex (tickets are handed out in submit order).  cvgs_queue_destroy
 * waits (at most 2 s) for the batches in flight, then retires the server.  Whether the host writes the ring straight into device
 * memory is decided without a fault (large-BAR attribute + /proc/self/maps + a read-back; CVGS_QUEUE_DIRECT=0 / 1 overrides).
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
 * gate is still closed wait there; other batches proceed.  A gate that stays closed for 10 s (CVGS_QUEUE_GATE_TIMEOUT_MS) is reported
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
/* out[16], of the last RETIRED server, 100 MHz ticks / counts: feeder {rounds with copies, slots, load ticks, copy ticks},
 * monitor {scans, scan ticks, completions published, -}, worker 0 {tasks, find ticks, rows ticks, drain ticks, idle polls} */
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

/* ---- streaming copy ----------------------------------------------------------------------------
 * dst[0..bytes) <- src[0..bytes) with the CircularTensor's plane-copy kernel (non-temporal 16-byte accesses), asynchronous
 * on `stream`.  No reference equivalent: it is the on-box "streaming kernel" copy ceiling SURVEY.md 8(d) asks the
 * roofline fractions to be quoted against (bench.py), and a utility for callers that re-pack tensors.  The two
 * ranges must not overlap.                                                                                      */
int cvgs_stream_copy(void* dst, const void* src, size_t bytes, cvgs_stream_t stream);

/* ---- test / measurement aid (ABI 5) --------------------------------------------------------------
 * `blocks` workgroups of `threads` threads that hold their wave slots and `lds_bytes` of LDS each for `microseconds`, asynchronous on
 * `stream`: a stand-in for a foreign kernel that occupies part of the chip (the queue's residency / watchdog tests, bench.py's
 * coexistence leg).  No reference counterpart.                                                                         */
int cvgs_debug_occupy(int32_t blocks, int32_t threads, int32_t lds_bytes, double microseconds, cvgs_stream_t stream);
/* one wave that reads `word` (device, uncached-device or pinned host memory, 8-byte aligned) with system-scope loads for `microseconds`
 * (nap != 0: s_sleep between the loads): the access pattern of a resident server's polling, for tools/probes/ only.               */
int cvgs_debug_poll(const void* word, double microseconds, int32_t nap, cvgs_stream_t stream);

/* ---- profiling ranges (reference tests/nvtx.h PUSH_RANGE/POP_RANGE) -------------------------- */
void cvgs_range_push(const char* name);
void cvgs_range_pop(void);

#ifdef __cplusplus
}
#endif
#endif /* CVGS_HIP_H */
