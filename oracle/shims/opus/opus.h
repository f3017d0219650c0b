/* oracle/shims/opus/opus.h -- TEST INFRASTRUCTURE: declaration-only stand-in (libopus is an un-vendored dependency) so the
 * reference's radio.h parses; the oracle wrappers never encode audio. */
#ifndef ORACLE_SHIM_OPUS_H
#define ORACLE_SHIM_OPUS_H
#include <stdint.h>
typedef struct OpusEncoder OpusEncoder;
typedef int32_t opus_int32;
typedef int16_t opus_int16;
#define OPUS_OK 0
#define OPUS_AUTO (-1000)
#define OPUS_BITRATE_MAX (-1)
#define OPUS_APPLICATION_VOIP 2048
#define OPUS_APPLICATION_AUDIO 2049
#define OPUS_APPLICATION_RESTRICTED_LOWDELAY 2051
#define OPUS_SIGNAL_VOICE 3001
#define OPUS_SIGNAL_MUSIC 3002
#define OPUS_BANDWIDTH_NARROWBAND 1101
#define OPUS_BANDWIDTH_MEDIUMBAND 1102
#define OPUS_BANDWIDTH_WIDEBAND 1103
#define OPUS_BANDWIDTH_SUPERWIDEBAND 1104
#define OPUS_BANDWIDTH_FULLBAND 1105
OpusEncoder *opus_encoder_create(opus_int32 Fs, int channels, int application, int *error);
void opus_encoder_destroy(OpusEncoder *st);
int opus_encoder_ctl(OpusEncoder *st, int request, ...);
opus_int32 opus_encode_float(OpusEncoder *st, const float *pcm, int frame_size, unsigned char *data, opus_int32 max_data_bytes);
const char *opus_strerror(int error);
#endif
