/* host/wav_header.h — RIFF/WAVE header scan shared by the front ends (what read_wav_header does,
 * reference demod/mod/demod_mod.c:313-376: find "fmt ", read channels / rate / bits, find "data"). */
#ifndef SONDE_WAV_HEADER_H
#define SONDE_WAV_HEADER_H
#include <stdio.h>
#include <stdint.h>
#include <string.h>

static int wav_find4(FILE *fp, const char *tag) {
    char w[4] = { 0, 0, 0, 0 };
    int c;
    while ((c = fgetc(fp)) != EOF) {
        w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = (char)c;
        if (!memcmp(w, tag, 4)) return 0;
    }
    return -1;
}

static int wav_read_header(FILE *fp, int *sr, int *bits, int *nch) {
    unsigned char d[16];
    char t[4];
    if (fread(t, 1, 4, fp) < 4 || (strncmp(t, "RIFF", 4) && strncmp(t, "RF64", 4))) return -1;
    if (fread(t, 1, 4, fp) < 4) return -1;
    if (fread(t, 1, 4, fp) < 4 || strncmp(t, "WAVE", 4)) return -1;
    if (wav_find4(fp, "fmt ") < 0) return -1;
    if (fread(d, 1, 4, fp) < 4) return -1;          /* chunk size  */
    if (fread(d, 1, 2, fp) < 2) return -1;          /* format tag  */
    if (fread(d, 1, 2, fp) < 2) return -1;
    *nch = d[0] + (d[1] << 8);
    if (fread(d, 1, 4, fp) < 4) return -1;
    *sr = (int)((uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
    if (fread(d, 1, 4, fp) < 4) return -1;          /* byte rate   */
    if (fread(d, 1, 2, fp) < 2) return -1;          /* block align */
    if (fread(d, 1, 2, fp) < 2) return -1;
    *bits = d[0] + (d[1] << 8);
    if (wav_find4(fp, "data") < 0) return -1;
    if (fread(d, 1, 4, fp) < 4) return -1;
    fprintf(stderr, "sample_rate: %d\n", *sr);
    fprintf(stderr, "bits       : %d\n", *bits);
    fprintf(stderr, "channels   : %d\n", *nch);
    if (*bits != 8 && *bits != 16 && *bits != 32) return -1;
    if (*sr == 900001) *sr -= 1;                    /* demod_mod.c:369 (printed as read, used as 900000) */
    return 0;
}
#endif
