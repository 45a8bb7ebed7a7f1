/* oracle/csdr_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; PARITY UNPINNED).
 * Restates the three csdr stages of the wide-band receive pipe
 *   rtl_sdr | csdr convert_u8_f | csdr fir_decimate_cc 45 | csdr convert_f_s16 | fsk_demod -c
 * [call site: /root/reference/README.md:109,162; source UPSTREAM-RECALLED from ha7ilm/csdr
 *  libcsdr.c / csdr.c, cloned un-pinned by /root/reference/build_csdr.sh:4-5]. */
#ifndef PIRIP_CSDR_ORACLE_H
#define PIRIP_CSDR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void oracle_convert_u8_f(const unsigned char *input, float *output, int length);
void oracle_convert_f_s16(const float *input, short *output, int length);
int  oracle_firdes_filter_len(float transition_bw);
void oracle_firdes_lowpass_f_hamming(float *output, int length, float cutoff_rate);
/* complex in/out as interleaved float pairs; returns number of outputs written */
int  oracle_fir_decimate_cc(const float *input, float *output, int input_size, int decimation,
                            const float *taps, int taps_length);
/* Streaming driver restating the buffer loop of `csdr fir_decimate_cc D [tbw]` over a whole
 * buffer: block size the_bufsize (csdr "bigbufs" default 16384), taps padded with zeros to a
 * multiple of 4, overlap carried, trailing partial block dropped. Returns outputs written. */
long oracle_csdr_fir_decimate_stream(const float *in_c, long nsamp, float *out_c, long max_out,
                                     int decimation, float transition_bw, int bufsize);
#ifdef __cplusplus
}
#endif
#endif
