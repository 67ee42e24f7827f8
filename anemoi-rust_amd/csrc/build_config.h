// build_config.h -- what a build of the library contains.
//
// `make` builds THE PRODUCT: no switches.  Every A/B knob of rounds 1-4 sits at the value that measured best, and the
// kernels that were built, measured slower and kept as recorded negatives are not compiled in:
//     ANEMOI_MAC_MODE 1          32-bit-limb multiply-accumulate of mont32.h (k_mont_convert only): v_mad_u64_u32 carry-out
//     ANEMOI_ASM_MUL 1           squaring / multiplication from the generated assembly (mont29_asm_gen.h, coop2d_asm_gen.h)
//     ANEMOI_WIN 3               sliding-window bits of the lane-private S-box (3 LDS entries per lane)
//     ANEMOI_XDIGITS_ON 1        + the two extra window digits held in VGPRs
//     ANEMOI_WAVES 0             no forced register budget
//     ANEMOI_ALT_PRIO 1          wave priority alternating round by round (config 3: two wavefronts per SIMD)
//     ANEMOI_ARITH32_FIELDS 0    no field on the saturated 32-bit-limb lane arithmetic (4.3 M/s against 10.6)
//     ANEMOI_RADIX30_FIELDS 3    30-bit limbs for the 381/377-bit fields (its default lives in the generated field_consts_gen.h)
//     ANEMOI_COOP2D_FUSE 1, ANEMOI_COOP2D_NOVDST 0, ANEMOI_LDS_ENTRIES_9 0, ANEMOI_HOLD_INPUTS_MAX_NL 13
//     not compiled: k_jive2_coop<F, 64> -- one item per wavefront: the four-row fold product (11-limb fields; 86 issue
//                   slots against the two-row form's 79, Jubjub 1.073 against 0.960 ms) and rounds 1-2's one-element scan
//                   (15-limb fields) -- with its option `coop_max`.
// `make AB=1 [MAC_MODE=.. WIN=.. WAVES=.. ASM_MUL=.. ALT_PRIO=.. EXTRA=-D..] LIBNAME=..` builds a LABORATORY library: the
// switches are honoured, the negatives are compiled in and routable through `coop_max`; tools/ab_*.py load such builds
// side by side with the product, tests/test_gpu_parity.py forces the negatives when the loaded library is one.
#pragma once

#ifndef ANEMOI_AB_BUILD
#define ANEMOI_AB_BUILD 0
#endif

#if !ANEMOI_AB_BUILD && !defined(ANEMOI_BOUNDS_WALK)   // (the host walk of the bounds compiles without the assembly)
#if defined(ANEMOI_MAC_MODE) || defined(ANEMOI_ASM_MUL) || defined(ANEMOI_WIN) || defined(ANEMOI_XDIGITS_ON) ||            \
    defined(ANEMOI_WAVES) || defined(ANEMOI_ALT_PRIO) || defined(ANEMOI_ARITH32_FIELDS) ||                                   \
    defined(ANEMOI_COOP2D_FUSE) || defined(ANEMOI_COOP2D_NOVDST) || defined(ANEMOI_LDS_ENTRIES_9) ||                        \
    defined(ANEMOI_HOLD_INPUTS_MAX_NL)
#error "A/B switches are honoured by `make AB=1` only: the product build has none (csrc/build_config.h)"
#endif
#endif

#ifndef ANEMOI_MAC_MODE
#define ANEMOI_MAC_MODE 1
#endif
#ifndef ANEMOI_ASM_MUL
#define ANEMOI_ASM_MUL 1
#endif
#ifndef ANEMOI_WIN
#define ANEMOI_WIN 3
#endif
#ifndef ANEMOI_XDIGITS_ON
#define ANEMOI_XDIGITS_ON 1
#endif
#ifndef ANEMOI_WAVES
#define ANEMOI_WAVES 0
#endif
#ifndef ANEMOI_ALT_PRIO
#define ANEMOI_ALT_PRIO 1
#endif
#ifndef ANEMOI_ARITH32_FIELDS
#define ANEMOI_ARITH32_FIELDS 0
#endif
#ifndef ANEMOI_COOP2D_FUSE
#define ANEMOI_COOP2D_FUSE 1
#endif
#ifndef ANEMOI_COOP2D_NOVDST
#define ANEMOI_COOP2D_NOVDST 0
#endif
#ifndef ANEMOI_LDS_ENTRIES_9
#define ANEMOI_LDS_ENTRIES_9 0
#endif
#ifndef ANEMOI_HOLD_INPUTS_MAX_NL
#define ANEMOI_HOLD_INPUTS_MAX_NL 13
#endif
