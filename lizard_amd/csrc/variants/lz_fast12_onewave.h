// variants/lz_fast12_onewave.h — levels 10 / 30 in the one-wave-per-block form of rounds 1-2 (superseded by the producer / consumer
// kernel lz_fast12_split_kernel in round 3).  Compiled only into -DLZ_FAST12_SPLIT=0 tuning builds (lz_kernels.h includes this file
// under that switch); not part of the shipped kernels and not hashed into bench.py's kernel_source_sha16.
#pragma once
// levels 10 / 30, the one-wave-per-block form of rounds 1-2 (launched only by LZ_FAST12_SPLIT=0 builds, kept for side-by-side
// measurements — compiled into -DLZ_FAST12_SPLIT=0 variant builds only; the library runs lz_fast12_split_kernel below): fastSmall parser, 2^12-slot table + sequence ring (+ Huffman
// workspace).  MIXED: level 10 thirteen waves with the 12 KiB table in LDS; level 30 sixteen waves, eleven with the table in LDS
// and five with a u32-slot table in global memory; larger blocks run the all-LDS form (13 / 11 waves).
template <bool HUF, bool MIXED>
__global__ __launch_bounds__(64 * (MIXED ? (HUF ? LZ_WAVES_FAST_HUF : LZ_WAVES_FAST) : (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS)))
void lz_fast12_kernel(LzBatch a)
{
    if constexpr (MIXED)
        lz_wave_main<LZ_PARSER_FAST, 12, 0, HUF, (HUF ? LZ_WAVES_FAST_HUF : LZ_WAVES_FAST), (HUF && !LZ_HUF_POOL ? LZ_HUF_WS_WORDS : 1),
                     (HUF ? LZ_NLDS_FAST_HUF : LZ_NLDS_FAST), LZ_TABKIND_LDS, (HUF ? LZ_HUF_POOL : 0)>(a);
    else
        lz_wave_main<LZ_PARSER_FAST, 12, 0, HUF, (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS), (HUF ? LZ_HUF_WS_WORDS : 1),
                     (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS)>(a);
}
