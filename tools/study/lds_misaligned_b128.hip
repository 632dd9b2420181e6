// Does a 2-byte-aligned ds_read_b128 return the right 16 bytes on gfx950?  (study for det_softmax_cand: the compiler turned eight
// consecutive f16 reads at [lane * 81 + c] into ds_read_b128 at lane * 162 + 2 + 16 i, and one detection's score came out 58 ulp off.)
// Fills a workgroup's LDS with a known pattern through 16-byte stores of whole rows (as the kernel does), reads it back per lane with
// (a) ds_read_b128 at the misaligned addresses and (b) ds_read_u16, and counts lanes whose eight halves differ.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_misaligned_b128 lds_misaligned_b128.hip ; run: ./lds_misaligned_b128 [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned short u16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(192) void probe(unsigned long long* bad, unsigned seed, int mode) {
    __shared__ __attribute__((aligned(16))) u16 zh[192 * 81];
    const int tid = threadIdx.x;
    // pattern: value at half index h of workgroup g = hash(h, g, seed) (never the same in neighbouring halves)
    const unsigned g = blockIdx.x * 2654435761u + seed;
    for (int h = tid; h < 192 * 81; h += 192) zh[h] = (u16)((h * 40503u + g) >> 7);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(zh) + tid * 162;  // LDS byte address of this lane's 81 halves
    unsigned long long wrong = 0;
    for (int i = 0; i < 10; ++i) {
        u32x4 v;
        const unsigned a = base + 2 + 16 * i;
        if (mode == 0) {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        } else {  // several in flight behind one wait, as compiled code has them
            u32x4 v2;
            const unsigned a2 = base + 2 + 16 * ((i + 5) % 10);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(1)\n\ts_nop 0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v), "=&v"(v2) : "v"(a), "v"(a2) : "memory");
        }
        for (int e = 0; e < 8; ++e) {
            const int h = tid * 81 + 1 + 8 * i + e;
            const u16 want = (u16)((h * 40503u + g) >> 7);
            const u16 got = (u16)(v[e >> 1] >> (16 * (e & 1)));
            wrong += got != want;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

// mode 2: seven misaligned reads in flight; each destination is copied right after the partial lgkmcnt wait that covers it (the
// destination registers start as a sentinel: a copy taken before the data arrived shows it)
__global__ __launch_bounds__(192) void probe7(unsigned long long* bad, unsigned seed) {
    __shared__ __attribute__((aligned(16))) u16 zh[192 * 81];
    const int tid = threadIdx.x;
    const unsigned g = blockIdx.x * 2654435761u + seed;
    for (int h = tid; h < 192 * 81; h += 192) zh[h] = (u16)((h * 40503u + g) >> 7);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(zh) + tid * 162;
    unsigned out[28];
    asm volatile("ds_read_b128 v[100:103], %0 offset:2\n\tds_read_b128 v[104:107], %0 offset:18\n\tds_read_b128 v[108:111], %0 offset:34\n\tds_read_b128 v[112:115], %0 offset:50\n\tds_read_b128 v[116:119], %0 offset:66\n\tds_read_b128 v[120:123], %0 offset:82\n\tds_read_b128 v[124:127], %0 offset:98\n\ts_waitcnt lgkmcnt(6)\n\tv_mov_b32 v130, v100\n\tv_mov_b32 v131, v101\n\tv_mov_b32 v132, v102\n\tv_mov_b32 v133, v103\n\ts_waitcnt lgkmcnt(5)\n\tv_mov_b32 v134, v104\n\tv_mov_b32 v135, v105\n\tv_mov_b32 v136, v106\n\tv_mov_b32 v137, v107\n\ts_waitcnt lgkmcnt(4)\n\tv_mov_b32 v138, v108\n\tv_mov_b32 v139, v109\n\tv_mov_b32 v140, v110\n\tv_mov_b32 v141, v111\n\ts_waitcnt lgkmcnt(3)\n\tv_mov_b32 v142, v112\n\tv_mov_b32 v143, v113\n\tv_mov_b32 v144, v114\n\tv_mov_b32 v145, v115\n\ts_waitcnt lgkmcnt(2)\n\tv_mov_b32 v146, v116\n\tv_mov_b32 v147, v117\n\tv_mov_b32 v148, v118\n\tv_mov_b32 v149, v119\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32 v150, v120\n\tv_mov_b32 v151, v121\n\tv_mov_b32 v152, v122\n\tv_mov_b32 v153, v123\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 v154, v124\n\tv_mov_b32 v155, v125\n\tv_mov_b32 v156, v126\n\tv_mov_b32 v157, v127" :: "v"(base) : "memory", "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157");
    asm volatile("v_mov_b32 %0, v130" : "=v"(out[0]));
    asm volatile("v_mov_b32 %0, v131" : "=v"(out[1]));
    asm volatile("v_mov_b32 %0, v132" : "=v"(out[2]));
    asm volatile("v_mov_b32 %0, v133" : "=v"(out[3]));
    asm volatile("v_mov_b32 %0, v134" : "=v"(out[4]));
    asm volatile("v_mov_b32 %0, v135" : "=v"(out[5]));
    asm volatile("v_mov_b32 %0, v136" : "=v"(out[6]));
    asm volatile("v_mov_b32 %0, v137" : "=v"(out[7]));
    asm volatile("v_mov_b32 %0, v138" : "=v"(out[8]));
    asm volatile("v_mov_b32 %0, v139" : "=v"(out[9]));
    asm volatile("v_mov_b32 %0, v140" : "=v"(out[10]));
    asm volatile("v_mov_b32 %0, v141" : "=v"(out[11]));
    asm volatile("v_mov_b32 %0, v142" : "=v"(out[12]));
    asm volatile("v_mov_b32 %0, v143" : "=v"(out[13]));
    asm volatile("v_mov_b32 %0, v144" : "=v"(out[14]));
    asm volatile("v_mov_b32 %0, v145" : "=v"(out[15]));
    asm volatile("v_mov_b32 %0, v146" : "=v"(out[16]));
    asm volatile("v_mov_b32 %0, v147" : "=v"(out[17]));
    asm volatile("v_mov_b32 %0, v148" : "=v"(out[18]));
    asm volatile("v_mov_b32 %0, v149" : "=v"(out[19]));
    asm volatile("v_mov_b32 %0, v150" : "=v"(out[20]));
    asm volatile("v_mov_b32 %0, v151" : "=v"(out[21]));
    asm volatile("v_mov_b32 %0, v152" : "=v"(out[22]));
    asm volatile("v_mov_b32 %0, v153" : "=v"(out[23]));
    asm volatile("v_mov_b32 %0, v154" : "=v"(out[24]));
    asm volatile("v_mov_b32 %0, v155" : "=v"(out[25]));
    asm volatile("v_mov_b32 %0, v156" : "=v"(out[26]));
    asm volatile("v_mov_b32 %0, v157" : "=v"(out[27]));
    unsigned long long wrong = 0;
    for (int q = 0; q < 56; ++q) {
        const int h = tid * 81 + 1 + q;
        const u16 want = (u16)((h * 40503u + g) >> 7);
        const u16 got = (u16)(out[q >> 1] >> (16 * (q & 1)));
        wrong += got != want;
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    unsigned long long* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 8);
        for (int l = 0; l < launches; ++l) probe<<<64 * 101, 192>>>(bad, (unsigned)l * 7919u + 1u, mode);
        unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %d launches x %d workgroups x 192 lanes x 80 halves: %llu wrong halves\n", mode, launches, 64 * 101, h);
    }
    hipMemset(bad, 0, 8);
    for (int l = 0; l < launches; ++l) probe7<<<64 * 101, 192>>>(bad, (unsigned)l * 7919u + 1u);
    unsigned long long h7 = 0; hipMemcpy(&h7, bad, 8, hipMemcpyDeviceToHost);
    printf("mode 2 (7 in flight, partial waits): %llu wrong halves\n", h7);
    return 0;
}
