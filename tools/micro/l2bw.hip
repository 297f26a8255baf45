// Microbenchmark: aggregate L2 -> CU read bandwidth on gfx950 for three access paths, each workgroup
// re-reading a small L2-resident slab: (0) LDS-DMA (buffer_load ... lds), (1) global_load_dwordx4 into
// registers, (2) global_load into registers followed by ds_write_b128.  Prints TB/s per path.  A fourth pass reads
// MFMA A-operand fragments of a [rows][row_bytes] weight slab straight into registers, with the waves of a workgroup
// on the same rows or on different ones.  Every buffer descriptor's num_records is the size of the allocation it
// covers (checked on the host): a descriptor longer than its allocation lets loads run past it, which faults.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE>
__global__ __launch_bounds__(256) void l2bw(const char* src, long long slab_bytes, int iters, int inflight, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned nrec = (unsigned)slab_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    // each wave instruction moves 1 KiB; a block pass = 4 KiB; walk the slab cyclically
    const unsigned mask = (unsigned)slab_bytes - 1;
    unsigned off = (unsigned)((blockIdx.x * 7919u * 4096u) & mask);
    i32x4 accv = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int k = 0; k < 8; ++k) {
            const unsigned so = __builtin_amdgcn_readfirstlane((off + (unsigned)wave * 1024u) & mask);
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 8 + k) * 1024), 16,
                                                         (unsigned)lane * 16u, so, 0, 0);
            } else {
                i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, so, 0);
                if (MODE == 2) *(i32x4*)(smem + (wave * 8 + k) * 1024 + lane * 16) = v;
                else accv += v;
            }
            off += 4096u;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 2) { __syncthreads(); accv = *(i32x4*)(smem + tid * 16); }
    if (MODE == 0) { __syncthreads(); accv = *(i32x4*)(smem + tid * 16); }
    if (accv[0] == 0x12345678 && accv[1] == 77) sink[0] = accv[2];
}

// A-operand fragment pattern: lane (fr = lane & 15, fq = lane >> 4) reads 16 bytes of row (16 j + fr) at
// column byte (k + 16 fq) of a [rows][row_bytes] weight slab; SHARE = waves of a workgroup read the same rows.
template <int SHARE>
__global__ __launch_bounds__(256) void fragbw(const char* src, int rows, int row_bytes, int iters, int* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(rows * row_bytes), 0x00020000);
    const unsigned vo = (unsigned)((lane & 15) * row_bytes + (lane >> 4) * 16);
    const int rowblocks = rows / 64;                       // 64-row groups (4 fragments) per slab
    int rb = SHARE ? (blockIdx.x % rowblocks) : ((blockIdx.x + wave) % rowblocks);
    i32x4 accv = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < row_bytes; k += 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((rb * 64 + j * 16) * row_bytes + k));
                accv += __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
            }
        }
    }
    if (accv[0] == 0x12345678 && accv[1] == 77) sink[0] = accv[2];
}

int main(int argc, char** argv) {
    const long long slab = (argc > 1 ? atoll(argv[1]) : 1) << 20;   // MiB, power of two
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;
    char* d; int* sink;
    if (slab <= 0 || (slab & (slab - 1)) || slab > (1LL << 31)) { printf("slab must be a power of two of MiB, at most 2048\n"); return 1; }
    if (hipMalloc(&d, slab) != hipSuccess || hipMemset(d, 1, slab) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("FAILED (alloc)\n"); return 1; }
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(l2bw<0>, dim3(blocks), dim3(256), 32768, 0, d, slab, iters, 0, sink);
            if (mode == 1) hipLaunchKernelGGL(l2bw<1>, dim3(blocks), dim3(256), 32768, 0, d, slab, iters, 0, sink);
            if (mode == 2) hipLaunchKernelGGL(l2bw<2>, dim3(blocks), dim3(256), 32768, 0, d, slab, iters, 0, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)blocks * iters * 8 * 4096;
            if (rep) printf("slab %lld MiB, %d blocks, mode %d (%s): %.3f ms, %.2f TB/s\n", slab >> 20, blocks, mode,
                            mode == 0 ? "lds-dma" : mode == 1 ? "load->vgpr" : "load->vgpr->lds", ms, bytes / ms * 1e-9);
        }
    }
    {
        const int rows = 512, row_bytes = 4608;           // 2.25 MiB slab: fits the 4 MiB L2 of one XCD
        // the fragment passes get a buffer of their own, exactly as large as the descriptor the kernel builds over it
        const size_t frag_bytes = (size_t)rows * row_bytes;
        char* fd = nullptr;
        if (hipMalloc(&fd, frag_bytes) != hipSuccess || hipMemset(fd, 1, frag_bytes) != hipSuccess) { printf("FAILED (alloc)\n"); return 1; }
        for (int share = 0; share < 2; ++share)
            for (int rep = 0; rep < 2; ++rep) {
                const int it2 = 40;
                hipEventRecord(e0);
                if (share) hipLaunchKernelGGL(fragbw<1>, dim3(blocks), dim3(256), 0, 0, fd, rows, row_bytes, it2, sink);
                else hipLaunchKernelGGL(fragbw<0>, dim3(blocks), dim3(256), 0, 0, fd, rows, row_bytes, it2, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)blocks * 4 * it2 * (row_bytes / 64) * 4 * 1024.0;
                if (rep) printf("fragment pattern, waves %s: %.3f ms, %.2f TB/s at the TA (%.1f B/clk/CU)\n",
                                share ? "share rows" : "distinct rows", ms, bytes / ms * 1e-9, bytes / ms * 1e-9 * 1e12 / 256 / 2.4e9);
            }
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    return 0;
}
