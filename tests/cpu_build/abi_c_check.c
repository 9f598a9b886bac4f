/* The boundary is a C ABI: include/vsp.h must compile as plain C99 and every entry point must link from a C program.
 * Host-only calls run here without a GPU; with a GPU the program also performs 2*G through vsp_msm_g1. */
#include <stdio.h>
#include <string.h>
#include "../../include/vsp.h"

int main(void) {
    static const uint64_t G1[12] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL,
                                    0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
    uint8_t enc[48];
    uint64_t back[12];
    int inf = 1;
    if (vsp_g1_compress(G1, enc) != VSP_OK) return 1;
    if (vsp_g1_decompress(enc, 1, back, &inf) != VSP_OK || inf || memcmp(back, G1, sizeof G1)) return 2;
    printf("generator round trip ok, first byte %02x\n", enc[0]);
    vsp_ctx *ctx = vsp_create(0);
    if (!ctx) { printf("no GPU: host-only part done\n"); return 77; }
    uint64_t bases[24], scalars[8] = {1, 0, 0, 0, 1, 0, 0, 0}, out[12];
    memcpy(bases, G1, sizeof G1); memcpy(bases + 12, G1, sizeof G1);
    int rc = vsp_msm_g1(ctx, bases, scalars, 2, out, &inf);
    printf("vsp_msm_g1 rc=%d inf=%d 2G.x[0]=%016llx\n", rc, inf, (unsigned long long)out[0]);
    vsp_destroy(ctx);
    return rc == VSP_OK && !inf ? 0 : 3;
}
