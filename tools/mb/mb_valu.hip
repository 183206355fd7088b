// micro-benchmark: how fast does gfx950 issue wave64 integer vector instructions?
//
// bench.py prices the entropy stage against a vector-issue peak; round 3 assumed 4 cycles per wave64 instruction per
// SIMD (256 CUs x 4 SIMDs x 2.4 GHz / 4 = 614 G wave-instructions/s), MI355X_MICROARCH.md says a wave64 VALU
// instruction takes 2 passes over a 32-wide SIMD when other waves fill the gaps.  This measures it: every kernel runs
// `ITER` iterations of `UNROLL` x 8 independent instructions of one kind (eight accumulator registers, so a wave alone
// never waits for its own result either), at 1, 2, 4 and 8 waves per SIMD on all CUs, and reports wave-instructions
// per second for the whole chip, cycles per instruction per SIMD (from the shader clock the kernel reads itself:
// s_memtime against the 100 MHz s_memrealtime), and the same for two-kind mixes (VALU + SALU from different waves,
// VALU + LDS).  Output: human-readable lines and one JSON object (profiles/r04_valu_peak.json is this program's output).
//
//   hipcc -O3 --offload-arch=gfx950 -o mb_valu mb_valu.hip && ./mb_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// kinds of the form "one instruction on an in/out vector register %0 and a constant vector register %1"
#define SIMPLE_KINDS(X)                                              \
	X(ADD_U32, "v_add_u32 %0, %0, %1")                               \
	X(SUB_U32, "v_sub_u32 %0, %0, %1")                               \
	X(MOV_B32, "v_mov_b32 %0, %1")                                   \
	X(AND_B32, "v_and_b32 %0, %0, %1")                               \
	X(OR_B32, "v_or_b32 %0, %0, %1")                                 \
	X(XOR_B32, "v_xor_b32 %0, %0, %1")                               \
	X(NOT_B32, "v_not_b32 %0, %0")                                   \
	X(LSHL_B32, "v_lshlrev_b32 %0, 1, %0")                           \
	X(LSHR_B32, "v_lshrrev_b32 %0, 1, %0")                           \
	X(ASHR_I32, "v_ashrrev_i32 %0, 1, %0")                           \
	X(LSHL_V_B32, "v_lshlrev_b32 %0, %1, %0")                        \
	X(MIN_U32, "v_min_u32 %0, %0, %1")                               \
	X(MAX_I32, "v_max_i32 %0, %0, %1")                               \
	X(ADD3_U32, "v_add3_u32 %0, %0, %1, %1")                         \
	X(LSHL_ADD_U32, "v_lshl_add_u32 %0, %0, 1, %1")                  \
	X(LSHL_OR_B32, "v_lshl_or_b32 %0, %0, 1, %1")                    \
	X(ADD_LSHL_U32, "v_add_lshl_u32 %0, %0, %1, 1")                  \
	X(AND_OR_B32, "v_and_or_b32 %0, %0, %1, %1")                     \
	X(OR3_B32, "v_or3_b32 %0, %0, %1, %1")                           \
	X(XAD_U32, "v_xad_u32 %0, %0, %1, %1")                           \
	X(BFE_U32, "v_bfe_u32 %0, %0, 1, 31")                            \
	X(BFI_B32, "v_bfi_b32 %0, %1, %0, %1")                           \
	X(ALIGNBIT, "v_alignbit_b32 %0, %0, %1, 7")                      \
	X(PERM_B32, "v_perm_b32 %0, %0, %1, %1")                         \
	X(BCNT, "v_bcnt_u32_b32 %0, %0, %1")                             \
	X(FFBL, "v_ffbl_b32 %0, %0")                                     \
	X(FFBH, "v_ffbh_u32 %0, %0")                                     \
	X(BFREV, "v_bfrev_b32 %0, %0")                                   \
	X(MAD_U32_U24, "v_mad_u32_u24 %0, %0, %1, %1")                   \
	X(MUL_U32_U24, "v_mul_u32_u24 %0, %0, %1")                       \
	X(MUL_LO_U32, "v_mul_lo_u32 %0, %0, %1")                         \
	X(SAD_U32, "v_sad_u32 %0, %0, %1, %1")                           \
	X(ADD_U16, "v_add_u16 %0, %0, %1")                               \
	X(PK_ADD_U16, "v_pk_add_u16 %0, %0, %1")                         \
	X(PK_SUB_I16, "v_pk_sub_i16 %0, %0, %1")                         \
	X(PK_LSHL_B16, "v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]")     \
	X(PK_ASHR_I16, "v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]")     \
	X(PK_MAX_I16, "v_pk_max_i16 %0, %0, %1")                         \
	X(ADD_F32, "v_add_f32 %0, %0, %1")                               \
	X(FMA_F32, "v_fma_f32 %0, %0, %1, %1")                           \
	X(CVT_F32_U32, "v_cvt_f32_u32 %0, %0")                       \
	X(DPP_ROW_SHR, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
	X(DPP_ADD_ROW_SHR, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
	X(DPP_ROW_BCAST, "v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf") \
	X(MBCNT_LO, "v_mbcnt_lo_u32_b32 %0, %1, %0")

enum Kind {
#define X(name, str) K_##name,
	SIMPLE_KINDS(X)
#undef X
	K_SIMPLE_END,
	K_READLANE = K_SIMPLE_END,
	K_READFIRSTLANE,
	K_CNDMASK,
	K_LSHL_B64,
	K_ADD_CO_PAIR,
	K_CMP_BALLOT,
	K_SALU_ADD,
	K_MIX_VALU_SALU,
	K_MIX_SHIFT_SALU,
	K_LDS_READ,
	K_COUNT
};
static const char *kind_name[K_COUNT] = {
#define X(name, str) str,
	SIMPLE_KINDS(X)
#undef X
	"v_readlane_b32", "v_readfirstlane_b32", "v_cndmask_b32 (sgpr pair)", "v_lshlrev_b64", "v_add_co_u32+v_addc_co_u32",
	"v_cmp_lt_u32 vcc + s_and_b64 (ballot)", "s_add_u32", "v_add_u32 + s_add_u32 (1:1)", "v_lshlrev_b32 + s_add_u32 (1:1)", "ds_read_b32 (dependent)",
};
// instructions per repetition (what the rate is divided by)
static int kind_insts(int k) { return k == K_ADD_CO_PAIR || k == K_CMP_BALLOT || k == K_MIX_VALU_SALU || k == K_MIX_SHIFT_SALU ? 2 : 1; }

constexpr int UNROLL = 8;     // x 8 registers = 64 instruction groups per iteration

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(unsigned *out, unsigned long long *clk, int iters, unsigned seed)
{
	unsigned r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u;
	unsigned long long w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7;
	unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
	const unsigned k = seed | 1u;
	const unsigned long long cond = 0x5555aaaa3333ccccull ^ seed;
	unsigned long long m0 = cond, m1 = cond + 1, m2 = cond + 2, m3 = cond + 3, m4 = cond + 4, m5 = cond + 5, m6 = cond + 6, m7 = cond + 7;
	__shared__ unsigned lds[256 * 8];
	if (KIND == K_LDS_READ) {
		for (int i = 0; i < 8; ++i)
			lds[threadIdx.x * 8 + i] = (threadIdx.x * 8 + i + 1) & 2047;
		__syncthreads();
	}
	unsigned long long t0 = 0, rt0 = 0;
	if (threadIdx.x == 0) {
		t0 = __builtin_amdgcn_s_memtime();
		rt0 = __builtin_amdgcn_s_memrealtime();
	}
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			if (KIND < K_SIMPLE_END) {
				switch (KIND) {
#define Y(i, str) asm volatile(str : "+v"(r##i) : "v"(k));
#define X(name, str) case K_##name: Y(0, str) Y(1, str) Y(2, str) Y(3, str) Y(4, str) Y(5, str) Y(6, str) Y(7, str) break;
				SIMPLE_KINDS(X)
#undef X
#undef Y
				default: break;
				}
			} else if (KIND == K_READLANE) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s##i) : "v"(r##i));
				REP8(X)
#undef X
			} else if (KIND == K_READFIRSTLANE) {
#define X(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s##i) : "v"(r##i));
				REP8(X)
#undef X
			} else if (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(k), "s"(cond));
				REP8(X)
#undef X
			} else if (KIND == K_LSHL_B64) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w##i));
				REP8(X)
#undef X
			} else if (KIND == K_ADD_CO_PAIR) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(r##i), "+v"(s##i) : "v"(k) : "vcc");
				// (s##i is a plain unsigned variable here: kept in a vector register by the "+v" constraint)
				REP8(X)
#undef X
			} else if (KIND == K_CMP_BALLOT) {
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\ts_and_b64 %0, vcc, %0" : "+s"(m##i) : "v"(r##i), "v"(k) : "vcc", "scc");
				REP8(X)
#undef X
			} else if (KIND == K_SALU_ADD) {
#define X(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s##i) : "s"(k) : "scc");
				REP8(X)
#undef X
			} else if (KIND == K_MIX_VALU_SALU) {
#define X(i) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(r##i), "+s"(s##i) : "v"(k), "s"(k) : "scc");
				REP8(X)
#undef X
			} else if (KIND == K_MIX_SHIFT_SALU) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0\n\ts_add_u32 %1, %1, %2" : "+v"(r##i), "+s"(s##i) : "s"(k) : "scc");
				REP8(X)
#undef X
			} else if (KIND == K_LDS_READ) {
#define X(i) r##i = lds[r##i & 2047];
				REP8(X)
#undef X
			}
		}
	}
	unsigned acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^
		(unsigned)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) ^ (unsigned)(m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7);
	if (threadIdx.x == 0) {
		const unsigned long long t1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
		if (blockIdx.x == 0) {
			clk[0] = t1 - t0;
			clk[1] = rt1 - rt0;
		}
	}
	if (acc == 0x12345678u)   // (never; keeps the results alive)
		out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
static void launch(int blocks, unsigned *out, unsigned long long *clk, int iters)
{
	hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 12345u);
}
typedef void (*launch_fn)(int, unsigned *, unsigned long long *, int);
template <int K>
struct Fill {
	static void go(launch_fn *t)
	{
		t[K] = launch<K>;
		Fill<K + 1>::go(t);
	}
};
template <>
struct Fill<K_COUNT> {
	static void go(launch_fn *) {}
};

int main(int argc, char **argv)
{
	const char *json_path = argc > 1 ? argv[1] : nullptr;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
	launch_fn table[K_COUNT];
	Fill<0>::go(table);
	unsigned *out;
	unsigned long long *clk, hclk[2];
	hipMalloc(&out, sizeof(unsigned) * 256 * cus * 16);
	hipMalloc(&clk, 16);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 8000;
	std::string json = "{\n \"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) + ", \"simds\": " + std::to_string(cus * 4) +
		",\n \"method\": \"tools/mb/mb_valu.hip: " + std::to_string(iters) + " iterations x " + std::to_string(UNROLL * 8) +
		" independent instructions per wave, workgroups of 4 waves, HIP events; cycles from s_memtime\",\n \"kinds\": {\n";
	double best_valu = 0, best_shift = 0;
	for (int kind = 0; kind < K_COUNT; ++kind) {
		json += std::string("  \"") + kind_name[kind] + "\": {";
		for (int wps : {1, 2, 4, 8}) {
			const int blocks = cus * wps;   // 4 waves per workgroup: one per SIMD, wps workgroups per CU
			table[kind](blocks, out, clk, 10);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			table[kind](blocks, out, clk, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
			const double insts = (double)blocks * 4 * iters * UNROLL * 8 * kind_insts(kind);
			const double rate = insts / (ms * 1e-3);
			const double mhz = hclk[1] ? (double)hclk[0] / ((double)hclk[1] / 100.0) : 0;   // s_memrealtime ticks at 100 MHz
			// cycles a SIMD spends per wave instruction: SIMD-cycles available / instructions issued
			const double cyc = (double)cus * 4 * (mhz * 1e6) * (ms * 1e-3) / insts;
			printf("%-30s %d waves/SIMD: %8.1f G wave-insts/s  %5.2f cycles/inst/SIMD  (shader clock %4.0f MHz, %6.3f ms)\n",
				kind_name[kind], wps, rate / 1e9, cyc, mhz, ms);
			char buf[256];
			snprintf(buf, sizeof buf, "%s\"%d\": {\"G_wave_insts_per_s\": %.1f, \"cycles_per_inst_per_simd\": %.3f, \"shader_mhz\": %.0f}",
				wps == 1 ? "" : ", ", wps, rate / 1e9, cyc, mhz);
			json += buf;
			if (kind == K_ADD_U32 && rate > best_valu)
				best_valu = rate;
			if (kind == K_LSHL_B32 && rate > best_shift)
				best_shift = rate;
		}
		json += kind + 1 < K_COUNT ? "},\n" : "}\n";
	}
	char buf[768];
	snprintf(buf, sizeof buf, " },\n \"fast_class_G_wave_insts_per_s\": %.1f,\n \"fast_class\": \"v_add_u32 and the other instructions that measure about 2.4 cycles per wave64 instruction per SIMD\",\n"
		" \"full_class_G_wave_insts_per_s\": %.1f,\n \"full_class\": \"v_lshlrev_b32 and the others at about 4.1 cycles: shifts, bit-field, permute, packed 16-bit, DPP moves, multiplies\"\n}\n",
		best_valu / 1e9, best_shift / 1e9);
	json += buf;
	printf("fast class (v_add_u32): %.1f, full class (v_lshlrev_b32): %.1f G wave-instructions/s\n", best_valu / 1e9, best_shift / 1e9);
	if (json_path) {
		FILE *f = fopen(json_path, "w");
		if (f) {
			fputs(json.c_str(), f);
			fclose(f);
		}
	}
	return 0;
}
