/*
 * stream8k.c -- BASELINE configs[4] from a plain-C host: a stream of 7680x4320 frames whose DCT32 and
 * SATD block batches are sharded across the GPUs of one node, through the node
 * part of include/x266hip.h.  One process drives every device (xHipNodeInit); the transfers are RCCL
 * send/recv groups over xGMI (or peer copies when RCCL is unavailable), pipelined against the kernels.
 * On a one-GPU box the same code runs with one rank and no transfer.
 *
 * It validates before it times: every frame of the pipelined, sharded run must equal, bit for bit, what
 * one device computes for the same frame with the plain batch calls.
 *
 *   usage: stream8k [n_ranks (0 = all visible devices)] [frames] [width] [height] [fused (1) | two launches (0)]      exit code 0 on success
 *          n_ranks > visible devices: ranks share devices round-robin (a test mode: RCCL refuses two ranks on one
 *          device, so the node uses its peer-copy transport -- the whole schedule still runs, with real transfers)
 *   prints one JSON line: frames/s, devices, transport, bit_exact
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/x266hip.h"

#define IN_RING X266_STREAM_IN_RING
#define OUT_RING X266_STREAM_OUT_RING

#define CHECK(call) do { int rc_ = (call); if (rc_ != X266HIP_OK) { fprintf(stderr, "%s failed: %d (%s | %s)\n", #call, rc_, \
    node ? xHipNodeLastError(node) : "", hip ? xHipLastError(hip) : ""); return 1; } } while (0)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
    int n_dev = argc > 1 ? atoi(argv[1]) : 0;
    const int frames = argc > 2 ? atoi(argv[2]) : 200;
    const int width = argc > 3 ? atoi(argv[3]) : 7680, height = argc > 4 ? atoi(argv[4]) : 4320;
    const int fused = argc > 5 ? atoi(argv[5]) : 1;
    x266hip_node *node = NULL;
    x266hip_ctx *hip = NULL;
    const int visible = xHipDeviceCount();
    if (visible <= 0) { fprintf(stderr, "no HIP device (this library has no CPU path)\n"); return 1; }
    if (n_dev <= 0) n_dev = visible;
    int devices[64];
    if (n_dev > 64) n_dev = 64;
    for (int i = 0; i < n_dev; i++) devices[i] = i % visible;
    if (xHipNodeInit(&node, devices, n_dev) != X266HIP_OK) { fprintf(stderr, "xHipNodeInit(%d devices) failed\n", n_dev); return 1; }
    hip = xHipNodeCtx(node, 0);                                    /* the root's context: frames live on its device */
    const int rccl = xHipNodeSelfTest(node) == X266HIP_OK;         /* ring send/recv + all-reduce over the node's communicators */
    if (!rccl) fprintf(stderr, "RCCL self-test did not pass (%s): peer-copy transport\n", xHipNodeLastError(node));
    if (!rccl && n_dev > 1) CHECK(xHipNodeSetOption(node, "transport", 1));   /* (already the node's fallback) */

    const size_t n_dct = (size_t)(width / 32) * (height / 32), n_satd = (size_t)(width / 8) * (height / 8);
    const size_t in_bytes[2] = {n_dct * 2048, n_satd * 128}, out_bytes[2] = {n_dct * 2048, n_satd * 4};
    void *d_in[IN_RING][2], *d_out[OUT_RING][2], *d_ref[IN_RING][2];
    for (int r = 0; r < IN_RING; r++)
        for (int l = 0; l < 2; l++) {
            CHECK(xHipMalloc(hip, &d_in[r][l], in_bytes[l]));
            CHECK(xHipMalloc(hip, &d_ref[r][l], out_bytes[l]));
            CHECK(xFillResidualDev(hip, (int16_t *)d_in[r][l], in_bytes[l] / 2, l ? 0x267 : 0x266, (uint64_t)r * 100000007u, NULL));
        }
    for (int r = 0; r < OUT_RING; r++)
        for (int l = 0; l < 2; l++) CHECK(xHipMalloc(hip, &d_out[r][l], out_bytes[l]));
    /* what ONE device computes for each distinct frame: the plain batch calls */
    for (int r = 0; r < IN_RING; r++) {
        CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_in[r][0], (int16_t *)d_ref[r][0], n_dct, NULL));
        CHECK(xSatd8x8BatchDev(hip, (const int16_t *)d_in[r][1], (uint32_t *)d_ref[r][1], n_satd, NULL));
    }
    CHECK(xHipStreamSync(hip, NULL));

    /* what the frame's kernels cost by themselves on one device: HIP events around back-to-back launches of the lane kernels */
    double kernel_us = 0.0;
    {
        void *e0 = NULL, *e1 = NULL;
        double ms = 0.0;
        const int reps = 200;
        CHECK(xHipEventCreate(hip, &e0));
        CHECK(xHipEventCreate(hip, &e1));
        for (int pass = 0; pass < 2; pass++) {                      /* pass 0 warms the clocks */
            CHECK(xHipEventRecord(hip, e0, NULL));
            for (int i = 0; i < reps; i++) {
                if (fused) CHECK(xDct32SatdFrameDev(hip, (const int16_t *)d_in[i % IN_RING][0], (int16_t *)d_out[i % OUT_RING][0], n_dct,
                                                    (const int16_t *)d_in[i % IN_RING][1], (uint32_t *)d_out[i % OUT_RING][1], n_satd, NULL));
                else {
                    CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_in[i % IN_RING][0], (int16_t *)d_out[i % OUT_RING][0], n_dct, NULL));
                    CHECK(xSatd8x8BatchDev(hip, (const int16_t *)d_in[i % IN_RING][1], (uint32_t *)d_out[i % OUT_RING][1], n_satd, NULL));
                }
            }
            CHECK(xHipEventRecord(hip, e1, NULL));
            CHECK(xHipEventElapsedMs(hip, e0, e1, &ms));
        }
        kernel_us = ms * 1e3 / reps;
        xHipEventDestroy(hip, e0);
        xHipEventDestroy(hip, e1);
    }

    x266hip_nstream *st = NULL;
    CHECK(xHipNodeSetOption(node, "fused_frame_lanes", fused));
    CHECK(xNodeFrameStreamCreate(node, width, height, &st));
    char *got = malloc(out_bytes[0] > out_bytes[1] ? out_bytes[0] : out_bytes[1]);
    char *want = malloc(out_bytes[0] > out_bytes[1] ? out_bytes[0] : out_bytes[1]);
    int exact = 1;
    /* validation pass: OUT_RING + 2 frames, each checked as soon as its ticket is complete */
    const int n_val = OUT_RING + 2;
    long ticket[OUT_RING + 2];
    for (int f = 0; f < n_val + 2; f++) {
        if (f < n_val) {
            const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
            void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
            CHECK(xNodeStreamPush(st, in, out, NULL, NULL, &ticket[f]));
        } else if (f == n_val) {
            CHECK(xNodeStreamFlush(st));                            /* the two draining steps */
        }
        const int g = f - 2;                                        /* frame whose results have just travelled */
        if (g >= 0 && g < n_val) {
            CHECK(xNodeStreamWait(st, ticket[g]));
            for (int l = 0; l < 2; l++) {
                CHECK(xHipMemcpyD2H(hip, got, d_out[g % OUT_RING][l], out_bytes[l]));
                CHECK(xHipMemcpyD2H(hip, want, d_ref[g % IN_RING][l], out_bytes[l]));
                if (memcmp(got, want, out_bytes[l])) { exact = 0; fprintf(stderr, "frame %d lane %d differs from the single-device result\n", g, l); }
            }
        }
    }
    /* timed pass */
    {                                                               /* warm-up: 0.1 s of frames -- the validation pass above idles the device
                                                                       between copies, and the chip needs ~50 ms of load to reach its steady clocks */
        const double tw = now_s();
        for (int f = 0; f < 8 || now_s() - tw < 0.1; f++) {
            const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
            void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
            CHECK(xNodeStreamPush(st, in, out, NULL, NULL, NULL));
        }
        CHECK(xNodeStreamFlush(st));
    }
    const double t0 = now_s();
    for (int f = 0; f < frames; f++) {
        const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
        void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
        /* the inputs are resident: "produced" on the frame's own slot stream, so the push needs no producer event */
        CHECK(xNodeStreamPush(st, in, out, NULL, xNodeStreamNextSlotStream(st), NULL));
    }
    CHECK(xNodeStreamFlush(st));
    const double dt = now_s() - t0;
    /* the last frame once more, after the timed run */
    for (int l = 0; l < 2; l++) {
        CHECK(xHipMemcpyD2H(hip, got, d_out[(frames - 1) % OUT_RING][l], out_bytes[l]));
        CHECK(xHipMemcpyD2H(hip, want, d_ref[(frames - 1) % IN_RING][l], out_bytes[l]));
        if (memcmp(got, want, out_bytes[l])) { exact = 0; fprintf(stderr, "last timed frame, lane %d differs\n", l); }
    }
    printf("{\"workload\": \"%dx%d frame stream: %zu DCT32 + %zu SATD blocks per frame\", \"ranks\": %d, \"visible_devices\": %d, \"transport\": \"%s\", "
           "\"frames\": %d, \"frames_per_s\": %.1f, \"ms_per_frame\": %.4f, \"kernel_us_per_frame\": %.2f, \"launches_per_frame\": %d, "
           "\"kernel_share_of_frame_time\": %.3f, \"dct32_blocks_per_s\": %.4e, \"satd8x8_blocks_per_s\": %.4e, "
           "\"bit_exact_vs_single_device\": %s}\n",
           width, height, n_dct, n_satd, n_dev, visible, n_dev == 1 ? "none (one rank)" : rccl ? "rccl send/recv groups" : "hipMemcpyPeerAsync",
           frames, frames / dt, dt / frames * 1e3, kernel_us, fused ? 1 : 2, kernel_us * 1e-6 / (dt / frames),
           n_dct * frames / dt, n_satd * frames / dt, exact ? "true" : "false");
    free(got); free(want);
    xNodeStreamFree(st);
    for (int r = 0; r < IN_RING; r++) for (int l = 0; l < 2; l++) { xHipFree(hip, d_in[r][l]); xHipFree(hip, d_ref[r][l]); }
    for (int r = 0; r < OUT_RING; r++) for (int l = 0; l < 2; l++) xHipFree(hip, d_out[r][l]);
    xHipNodeFree(node);
    return exact ? 0 : 2;
}
