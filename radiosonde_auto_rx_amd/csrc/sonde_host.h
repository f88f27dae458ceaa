// sonde_host.h — internal host-side declarations of libsonde_hip (design math, RS41 framing/ECC).
#ifndef SONDE_HOST_H
#define SONDE_HOST_H
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

namespace sonde {

struct Decimator { int if_sr = 0, decM = 1; std::vector<float> taps; };

std::vector<float> design_lowpass(float f, int taps);
Decimator design_decimator(int sr_base, bool if_min);
Decimator design_decimator_scan(int sr_base, bool if_min, float set_lpIQ);
Decimator design_decimator_if(int sr_base, int if_target, bool narrow);
struct Mixer { double f0 = 0; int lut_len = 1; };
Mixer design_mixer(double xlt_fq, int sr_base);
std::vector<float> design_match(const std::string &hdr, float sps, float bt);
void bit_window(int pos, int half, int symlen, float sps, uint32_t &q0, uint32_t &q1, double &mid);
void slice_range(uint32_t q0, uint32_t q1, double mid, float l, uint32_t &qa, uint32_t &qb);

// GF(2^8)/0x11D, alpha = 2
const uint8_t *gf_exp_table();   // 512 entries
const uint8_t *gf_log_table();   // 256 entries
int  rs255_syndromes(const uint8_t cw[255], uint8_t S[24]);     // returns 1 if any non-zero
int  rs255_decode_syn(uint8_t cw[255], const uint8_t S[24]);    // errors-only, Euclid; <0 on failure
int  rs255_decode(uint8_t cw[255]);
void rs255_encode(uint8_t cw[255]);
int  crc16(const uint8_t *p, int len);

// the reference's 8192-point transform (dft_raw with its float twiddle recurrence; sonde_scan.cpp): in place on 8192 (re, im) pairs,
// and the stage twiddle table the device kernels use (stage t at 2^t - 1 + j, (re, im) pairs)
void ref_dft_8192(std::vector<float> &re_im);
std::vector<float> ref_twiddle_table();

extern const char    kRs41Header[65];
extern const uint8_t kRs41HeaderBytes[8];
extern const uint8_t kRs41Mask[64];
int  rs41_frametype(const uint8_t *frame);
// rs41_ecc() of the reference for ecc levels 1/2; synd = device syndromes of the first pass or nullptr
int  rs41_ecc(uint8_t frame[518], int frmlen, int level, const uint8_t *synd);

extern const char kDfmRawHeader[33];
extern const char kM10RawHeader[33];
int  m10_checksum(const uint8_t *msg, int len);          // checkM10 (m10mod.c:594-628)
int  dfm_block(int level, const uint8_t *hb, const float *sb, int L, uint8_t *nib);

}  // namespace sonde
#endif
