/*
 * stream8k_ranks.c -- the frame stream of stream8k.c with ONE PROCESS PER GPU (the MPI / torchrun way of driving a
 * node), still plain C: the parent forks n_ranks - 1 children before anything touches HIP, rank 0 makes the RCCL id
 * (xHipNodeUniqueId) and hands its 128 bytes to the others through pipes, every rank opens its communicator with
 * xHipNodeInitRank and then makes the SAME sequence of xNodeStreamPush / Flush calls -- rank 0 with the frame
 * buffers, the others with NULL.  Rank 0 validates every frame of the first pass against the plain single-device
 * calls, times the rest and prints one JSON line.
 *
 *   usage: stream8k_ranks [n_ranks (0 = visible devices)] [frames] [width] [height]      exit code 0 on success
 *   rank r runs on device r % visible.  RCCL refuses two ranks on one device, so on a one-GPU box n_ranks > 1 only
 *   works with X266HIP_RCCL_LIB naming the tests' RCCL model (tests/rccl_model, multi-process mode).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "../include/x266hip.h"

#define IN_RING X266_STREAM_IN_RING
#define OUT_RING X266_STREAM_OUT_RING
#define N_VAL (OUT_RING + 2)
#define MAX_RANKS 64

static x266hip_node *node;
static x266hip_ctx *hip;
static int g_rank;

#define CHECK(call) do { int rc_ = (call); if (rc_ != X266HIP_OK) { fprintf(stderr, "rank %d: %s failed: %d (%s | %s)\n", g_rank, #call, rc_, \
    node ? xHipNodeLastError(node) : "", hip ? xHipLastError(hip) : ""); return 1; } } while (0)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* what every rank but the root does: the same steps, no buffers */
static int peer(x266hip_nstream *st, int frames)
{
    for (int f = 0; f < N_VAL; f++) CHECK(xNodeStreamPush(st, NULL, NULL, NULL, NULL, NULL));
    CHECK(xNodeStreamFlush(st));
    for (int f = 0; f < 8; f++) CHECK(xNodeStreamPush(st, NULL, NULL, NULL, NULL, NULL));
    CHECK(xNodeStreamFlush(st));
    for (int f = 0; f < frames; f++) CHECK(xNodeStreamPush(st, NULL, NULL, NULL, NULL, NULL));
    CHECK(xNodeStreamFlush(st));
    return 0;
}

static int root(x266hip_nstream *st, int frames, int width, int height, int n_ranks, int visible)
{
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
    for (int r = 0; r < IN_RING; r++) {                             /* what ONE device computes: the plain batch calls */
        CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_in[r][0], (int16_t *)d_ref[r][0], n_dct, NULL));
        CHECK(xSatd8x8BatchDev(hip, (const int16_t *)d_in[r][1], (uint32_t *)d_ref[r][1], n_satd, NULL));
    }
    CHECK(xHipStreamSync(hip, NULL));
    const size_t big = out_bytes[0] > out_bytes[1] ? out_bytes[0] : out_bytes[1];
    char *got = malloc(big), *want = malloc(big);
    int exact = 1;
    for (int f = 0; f < N_VAL; f++) {
        const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
        void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
        CHECK(xNodeStreamPush(st, in, out, NULL, NULL, NULL));
    }
    CHECK(xNodeStreamFlush(st));
    for (int f = N_VAL - OUT_RING; f < N_VAL; f++)                  /* the frames still in the output ring */
        for (int l = 0; l < 2; l++) {
            CHECK(xHipMemcpyD2H(hip, got, d_out[f % OUT_RING][l], out_bytes[l]));
            CHECK(xHipMemcpyD2H(hip, want, d_ref[f % IN_RING][l], out_bytes[l]));
            if (memcmp(got, want, out_bytes[l])) { exact = 0; fprintf(stderr, "frame %d lane %d differs from the single-device result\n", f, l); }
        }
    for (int f = 0; f < 8; f++) {
        const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
        void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
        CHECK(xNodeStreamPush(st, in, out, NULL, NULL, NULL));
    }
    CHECK(xNodeStreamFlush(st));
    const double t0 = now_s();
    for (int f = 0; f < frames; f++) {
        const void *in[2] = {d_in[f % IN_RING][0], d_in[f % IN_RING][1]};
        void *out[2] = {d_out[f % OUT_RING][0], d_out[f % OUT_RING][1]};
        CHECK(xNodeStreamPush(st, in, out, NULL, NULL, NULL));
    }
    CHECK(xNodeStreamFlush(st));
    const double dt = now_s() - t0;
    for (int l = 0; l < 2 && frames > 0; l++) {
        CHECK(xHipMemcpyD2H(hip, got, d_out[(frames - 1) % OUT_RING][l], out_bytes[l]));
        CHECK(xHipMemcpyD2H(hip, want, d_ref[(frames - 1) % IN_RING][l], out_bytes[l]));
        if (memcmp(got, want, out_bytes[l])) { exact = 0; fprintf(stderr, "last timed frame, lane %d differs\n", l); }
    }
    printf("{\"workload\": \"%dx%d frame stream: %zu DCT32 + %zu SATD blocks per frame\", \"processes\": %d, \"visible_devices\": %d, "
           "\"frames\": %d, \"frames_per_s\": %.1f, \"ms_per_frame\": %.4f, \"bit_exact_vs_single_device\": %s}\n",
           width, height, n_dct, n_satd, n_ranks, visible, frames, frames / dt, dt / frames * 1e3, exact ? "true" : "false");
    fflush(stdout);
    free(got); free(want);
    for (int r = 0; r < IN_RING; r++) for (int l = 0; l < 2; l++) { xHipFree(hip, d_in[r][l]); xHipFree(hip, d_ref[r][l]); }
    for (int r = 0; r < OUT_RING; r++) for (int l = 0; l < 2; l++) xHipFree(hip, d_out[r][l]);
    return exact ? 0 : 2;
}

int main(int argc, char **argv)
{
    int n_ranks = argc > 1 ? atoi(argv[1]) : 0;
    const int frames = argc > 2 ? atoi(argv[2]) : 200;
    const int width = argc > 3 ? atoi(argv[3]) : 7680, height = argc > 4 ? atoi(argv[4]) : 4320;
    if (n_ranks > MAX_RANKS) n_ranks = MAX_RANKS;
    /* the rank count must be known before the fork, and the fork must come before the first HIP call: ask a child */
    if (n_ranks <= 0) {
        int p[2];
        if (pipe(p)) return 1;
        const pid_t c = fork();
        if (c == 0) { const int v = xHipDeviceCount(); if (write(p[1], &v, sizeof v) != sizeof v) _exit(1); _exit(0); }
        if (read(p[0], &n_ranks, sizeof n_ranks) != sizeof n_ranks) n_ranks = 0;
        waitpid(c, NULL, 0);
        close(p[0]); close(p[1]);
        if (n_ranks <= 0) { fprintf(stderr, "no HIP device (this library has no CPU path)\n"); return 1; }
    }
    int id_pipe[MAX_RANKS][2];
    pid_t child[MAX_RANKS];
    for (int r = 1; r < n_ranks; r++) {
        if (pipe(id_pipe[r])) return 1;
        child[r] = fork();
        if (child[r] < 0) return 1;
        if (child[r] == 0) { g_rank = r; close(id_pipe[r][1]); break; }
        close(id_pipe[r][0]);
    }
    unsigned char id[X266HIP_NODE_ID_BYTES];
    if (g_rank == 0) {
        if (xHipNodeUniqueId(id) != X266HIP_OK) { fprintf(stderr, "xHipNodeUniqueId failed: RCCL could not be loaded\n"); return 1; }
        for (int r = 1; r < n_ranks; r++)
            if (write(id_pipe[r][1], id, sizeof id) != (ssize_t)sizeof id) return 1;
    } else if (read(id_pipe[g_rank][0], id, sizeof id) != (ssize_t)sizeof id) {
        return 1;
    }
    const int visible = xHipDeviceCount();
    if (visible <= 0) { fprintf(stderr, "rank %d: no HIP device\n", g_rank); return 1; }
    if (xHipNodeInitRank(&node, g_rank % visible, g_rank, n_ranks, id) != X266HIP_OK) {
        fprintf(stderr, "rank %d: xHipNodeInitRank failed\n", g_rank);
        return 1;
    }
    hip = xHipNodeCtx(node, 0);
    CHECK(xHipNodeSelfTest(node));
    x266hip_nstream *st = NULL;
    CHECK(xNodeFrameStreamCreate(node, width, height, &st));
    int rc = g_rank == 0 ? root(st, frames, width, height, n_ranks, visible) : peer(st, frames);
    xNodeStreamFree(st);
    xHipNodeFree(node);
    if (g_rank != 0) _exit(rc);
    for (int r = 1; r < n_ranks; r++) {
        int status = 0;
        waitpid(child[r], &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "rank %d ended with status %d\n", r, status); if (!rc) rc = 3; }
    }
    return rc;
}
