/* ka9q_filter_hip_ext.h -- what libka9q_filter_hip.so offers BEYOND ka9q-radio's filter.h (optional; a caller that only
 * knows filter.h never needs it).
 *
 * estimate_noise() (/root/reference/src/radio.c:1783-1866) reads master->fdomain[] on the host for every channel and
 * block; it is the only reader of the block spectrum outside filter.c, and the reason the drop-in copies 13 MB per
 * block back over PCIe by default (KA9Q_HIP_FDOMAIN=1).  The device runs the same function (kernel noise_est, pinned to
 * radio.c's own code to 1e-12): a radiod built with the three-line patch of INTEGRATION.md section 1 takes the estimate
 * from filter_hip_noise() and sets KA9Q_HIP_FDOMAIN=0.
 */
#ifndef KA9Q_FILTER_HIP_EXT_H
#define KA9Q_FILTER_HIP_EXT_H
#ifdef __cplusplus
extern "C" {
#endif
struct filter_in;
struct filter_out;
/* switch the device-side noise estimate on for every channel of `master` (samprate = Frontend.samprate, Hz; 0 = off).
   Returns 0, 1 if some channel sizes have no noise kernel (their slaves report NaN), -1 on bad arguments.
   Env KA9Q_HIP_NOISE_SAMPRATE=<Hz> does the same at create_filter_input time. */
int filter_hip_enable_noise(struct filter_in *master, double samprate);
/* N0 of the block the slave's last execute_filter_output() delivered (NaN: not available) */
double filter_hip_noise(struct filter_out const *slave);
/* returns once the device has finished every block handed to it so far */
int filter_hip_drain(struct filter_in *master);
/* blocks skipped by the front end because the device was ND blocks behind (only with KA9Q_HIP_INPUT_FULL=drop) */
unsigned long filter_hip_skipped_blocks(struct filter_in const *master);
/* Failure policy.  A failed device-side check or a HIP error while a block is being enqueued makes the drop-in replace its engine
   ONCE (every registered slave's response, shift, ISB / beam state, the notch list and the overlap history are carried over; the
   blocks lost on the way are zeros + block_drops for every slave, like a lapped block, /root/reference/src/filter.c:690-701); a second
   failure within 500 blocks, or a re-creation that fails, ends the process with EX_SOFTWARE so that the supervisor restarts it, which
   is what the reference does on a fatal error (/root/reference/src/radio.c:398, src/main.c:202).  Returns the number of recoveries so far. */
unsigned filter_hip_recoveries(struct filter_in const *master, unsigned *blocks_lost);
/* Sharding behind filter.h.  Env KA9Q_HIP_DEVICES="0,1,..." at create_filter_input time spreads the slaves of the master over the
   listed devices (one engine each; every device takes the block's samples from the same pinned host ring and transforms them
   itself; slaves are assigned in creation order, KA9Q_HIP_SHARD_CHANNELS -- default 1024 -- per device before the next one is
   used; a block is complete when every device has delivered).  The reference runs a thread per channel inside one process against
   one shared master (/root/reference/src/radio.c:996, src/filter.c:704-712); callers of filter.h see no difference.  Returns the
   number of devices; counts (may be NULL) receives the slaves living on each. */
int filter_hip_devices(struct filter_in const *master, int *counts, int max);
/* round 6: what the process must still do before the drop-in ends it for the supervisor (a second device failure, a re-creation that
 * fails, a device that completes nothing for KA9Q_HIP_WEDGED_MS).  The reference's fatal path shuts the front end down first
 * (src/main.c:197-201: Frontend.shutdown -- bias tee off, streaming stopped); radiod registers the equivalent here, e.g.
 *     static void hw_off(void) { if (Frontend.shutdown) Frontend.shutdown(&Frontend); }   ...   filter_hip_set_exit_hook(hw_off);
 * Called once, from the failing thread, before stdio is flushed and _exit(EX_SOFTWARE).  NULL removes it. */
void filter_hip_set_exit_hook(void (*hook)(void));
#ifdef __cplusplus
}
#endif
#endif
