// Micro-test (diagnostic): lane semantics of the cross-lane moves a register-level gamma broadcast would use.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int lane = threadIdx.x;
    int v = lane, r;
    // row_ror:12 / row_ror:4 on rows 2,3 only, bank masks
    r = __builtin_amdgcn_update_dpp(-1, v, 0x120 + 12, 0xc, 0x1, false); out[0 * 64 + lane] = r;    // row_ror:12, rows 2,3, bank 0
    r = __builtin_amdgcn_update_dpp(-1, v, 0x120 + 4, 0xc, 0x6, false); out[1 * 64 + lane] = r;     // row_ror:4, rows 2,3, banks 1,2
    r = __builtin_amdgcn_update_dpp(-1, v, 0x120 + 8, 0xc, 0xf, false); out[2 * 64 + lane] = r;     // row_ror:8, rows 2,3
    r = __builtin_amdgcn_update_dpp(-1, v, 0x150 + 5, 0xf, 0xf, false); out[3 * 64 + lane] = r;     // row_newbcast:5
    int a = lane, b = 100 + lane;
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[4 * 64 + lane] = s[0]; out[5 * 64 + lane] = s[1];
    r = __builtin_amdgcn_update_dpp(-1, v, 0xE4, 0xc, 0xf, false); out[6 * 64 + lane] = r;          // quad_perm [0,1,2,3] (identity), rows 2,3
}
int main()
{
    int *d, h[7 * 64];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[7] = {"row_ror:12 rows23 bank0", "row_ror:4 rows23 banks12", "row_ror:8 rows23", "row_newbcast:5", "permlane16_swap[0] (a=lane)", "permlane16_swap[1] (b=100+lane)", "quad_perm id rows23"};
    for (int t = 0; t < 7; ++t) {
        printf("%-32s", names[t]);
        for (int i = 0; i < 64; ++i) printf(" %d", h[t * 64 + i]);
        printf("\n");
    }
    return 0;
}
