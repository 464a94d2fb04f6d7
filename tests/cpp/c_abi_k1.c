/* c_abi_k1.c -- the drop-in boundary from PLAIN C (what a cgo / JNI / N-API binding does): the headline chain
 *   8 crops of a 1080p u8c3 frame -> bilinear resize 64x128 -> RGB<->BGR -> x0.3 -> -(1,4,3.2) -> /(3.2,0.6,11.8) -> NCHW fp32
 * described as ONE cvgs_chain_desc (include/cvgs_hip.h: plain structs, pointers and sizes), executed with cvgs_execute on a HIP stream,
 * and compared bit for bit with the CPU oracle running the SAME descriptor on host memory.
 *   make -C tests/cpp bin/c_abi_k1 && tests/cpp/bin/c_abi_k1   (gcc -std=c99: no C++ anywhere in this file; a TEST -- it links the oracle)  */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cvgs_hip.h"
#include "../../oracle/cvgs_oracle.h"

#define N 8
#define W 1920
#define H 1080
#define DW 64
#define DH 128

static uint64_t g_s = 0xC0FFEEull;
static uint64_t rnd(void) {
    g_s += 0x9E3779B97F4A7C15ull;
    uint64_t z = g_s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void describe(cvgs_chain_desc* d, cvgs_image2d* crops, const uint8_t* frame, size_t step, void* tensor, const int (*rects)[4]) {
    const float mul[3] = {0.3f, 0.3f, 0.3f}, sub[3] = {1.f, 4.f, 3.2f}, dv[3] = {3.2f, 0.6f, 11.8f};
    int i, c;
    memset(d, 0, sizeof(*d));
    d->struct_size = (uint32_t)sizeof(*d);
    for (i = 0; i < N; ++i) { /* a crop is a view: pointer to its first pixel, extent, the frame's pitch */
        crops[i].data = frame + (size_t)rects[i][1] * step + (size_t)rects[i][0] * 3;
        crops[i].width = rects[i][2];
        crops[i].height = rects[i][3];
        crops[i].step = (int32_t)step;
        crops[i].uv_offset = 0;
    }
    d->read.kind = CVGS_READ_RESIZE_LINEAR;
    d->read.src_type = CVGS_MAKETYPE(0 /* CV_8U */, 3);
    d->read.batch = d->read.used_planes = N;
    d->read.src = crops;
    d->read.dst_width = DW;
    d->read.dst_height = DH;
    d->read.aspect_ratio = CVGS_IGNORE_AR;
    d->n_ops = 4;
    d->ops[0].opcode = CVGS_OP_REORDER; /* cv::COLOR_RGB2BGR: out[c] = in[2 - c] */
    d->ops[0].aux = 2 | (1 << 2) | (0 << 4) | (3 << 6);
    d->ops[1].opcode = CVGS_OP_MUL;
    d->ops[2].opcode = CVGS_OP_SUB;
    d->ops[3].opcode = CVGS_OP_DIV;
    for (c = 0; c < 3; ++c) {
        d->ops[1].operand[c] = mul[c]; d->ops[1].operand_d[c] = mul[c];
        d->ops[2].operand[c] = sub[c]; d->ops[2].operand_d[c] = sub[c];
        d->ops[3].operand[c] = dv[c];  d->ops[3].operand_d[c] = dv[c];
    }
    d->write.kind = CVGS_WRITE_TENSOR_SPLIT;
    d->write.dst_type = CVGS_MAKETYPE(5 /* CV_32F */, 3);
    d->write.data = tensor;
    d->write.width = DW;
    d->write.height = DH;
    d->write.planes = N;
}

int main(void) {
    const size_t frame_bytes = (size_t)W * H * 3, tensor_bytes = (size_t)N * 3 * DW * DH * sizeof(float);
    uint8_t* h_frame = (uint8_t*)malloc(frame_bytes);
    float *h_ref = (float*)malloc(tensor_bytes), *h_got = (float*)malloc(tensor_bytes);
    void *d_frame = NULL, *d_tensor = NULL;
    hipStream_t stream;
    cvgs_chain_desc gpu, cpu;
    cvgs_image2d gpu_crops[N], cpu_crops[N];
    int rects[N][4], i, rc;
    char kernel[128];
    size_t k;

    for (k = 0; k + 8 <= frame_bytes; k += 8) { const uint64_t v = rnd(); memcpy(h_frame + k, &v, 8); }
    for (i = 0; i < N; ++i) {
        rects[i][2] = 16 + (int)(rnd() % 600);
        rects[i][3] = 16 + (int)(rnd() % 700);
        rects[i][0] = (int)(rnd() % (uint64_t)(W - rects[i][2] + 1));
        rects[i][1] = (int)(rnd() % (uint64_t)(H - rects[i][3] + 1));
    }
    if (cvgs_abi_version() != CVGS_ABI_VERSION) { fprintf(stderr, "libcvgs_hip.so speaks ABI %d, this header %d\n", cvgs_abi_version(), CVGS_ABI_VERSION); return 2; }
    if (hipMalloc(&d_frame, frame_bytes) != hipSuccess || hipMalloc(&d_tensor, tensor_bytes) != hipSuccess ||
        hipStreamCreate(&stream) != hipSuccess || hipMemcpy(d_frame, h_frame, frame_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "no HIP device / allocation failed\n");
        return 2;
    }
    describe(&gpu, gpu_crops, (const uint8_t*)d_frame, (size_t)W * 3, d_tensor, (const int (*)[4])rects);
    describe(&cpu, cpu_crops, h_frame, (size_t)W * 3, h_ref, (const int (*)[4])rects);

    rc = cvgs_execute(&gpu, stream); /* ONE kernel, asynchronous on `stream`, no synchronisation inside */
    if (rc != CVGS_OK) { fprintf(stderr, "cvgs_execute: %d (%s)\n", rc, cvgs_last_error()); return 1; }
    if (hipMemcpyAsync(h_got, d_tensor, tensor_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return 1;
    rc = oracle_execute(&cpu); /* the checker: the same descriptor on host pointers */
    if (rc != 0) { fprintf(stderr, "oracle_execute: %d\n", rc); return 1; }
    cvgs_kernel_name(&gpu, kernel, sizeof(kernel));
    rc = memcmp(h_got, h_ref, tensor_bytes);
    printf("%s: %d crops -> [%d,3,%d,%d] fp32, %s the CPU oracle\n", kernel, N, N, DH, DW, rc == 0 ? "bit-identical to" : "DIFFERENT from");
    (void)hipFree(d_frame);
    (void)hipFree(d_tensor);
    (void)hipStreamDestroy(stream);
    free(h_frame); free(h_ref); free(h_got);
    return rc == 0 ? 0 : 1;
}
