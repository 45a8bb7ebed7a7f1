/* oracle/kiss_fft_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned).
 * Float32 complex FFT restating kiss_fft as used by codec2's FSK frequency estimator
 * [UPSTREAM-RECALLED: codec2 src/kiss_fft.h]. */
#ifndef PIRIP_KISS_FFT_ORACLE_H
#define PIRIP_KISS_FFT_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { float r; float i; } kiss_fft_oracle_cpx;
typedef struct kiss_fft_oracle_state *kiss_fft_oracle_cfg;
kiss_fft_oracle_cfg kiss_fft_oracle_alloc(int nfft, int inverse_fft);
void kiss_fft_oracle_free(kiss_fft_oracle_cfg st);
void kiss_fft_oracle(kiss_fft_oracle_cfg st, const kiss_fft_oracle_cpx *fin, kiss_fft_oracle_cpx *fout);
const kiss_fft_oracle_cpx *kiss_fft_oracle_twiddles(kiss_fft_oracle_cfg st);
#ifdef __cplusplus
}
#endif
#endif
