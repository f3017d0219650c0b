/* oracle/ref_radio_wrap.c -- TEST INFRASTRUCTURE.  Pins the restated estimate_noise() (chz_oracle.c:chzo_estimate_noise,
 * SURVEY 8f rank 2) to the REFERENCE'S OWN CODE: this translation unit is the reference's src/radio.c, included
 * unmodified from where it lies, plus three exported wrappers around its static functions
 *     quickselect()  src/radio.c:1724-1757      quantile()  src/radio.c:1759-1775
 *     estimate_noise()  src/radio.c:1783-1866
 * Everything else radio.c defines is hidden and discarded at link time (-fvisibility=hidden -ffunction-sections
 * -Wl,--gc-sections), so none of radiod's other dependencies is needed.  Never copied into the repo; built by
 * oracle/Makefile only where /root/reference exists, into oracle/_ref/. */
#include "radio.c"

#define EXPORT __attribute__((visibility("default")))

struct frontend Frontend;          /* defined in the reference's main.c; estimate_noise reads Frontend.samprate only */

EXPORT double refradio_quickselect(double *a, int n, int k) { return quickselect(a, 0, n - 1, k); }
EXPORT double refradio_quantile(double *a, int n, double p) { return quantile(a, n, p); }

/* estimate_noise(chan, shift) on a caller-supplied master spectrum: builds the minimal chan/slave/master triple the
 * function dereferences (chan->filter.out.{bins,master,next_jobnum}, master->{in_type,bins,fdomain[]}) */
EXPORT double refradio_estimate_noise(const float complex *spectrum, int master_bins, int in_type, int slave_bins,
                                      int shift, double samprate) {
  static struct filter_in master;     /* zeroed; only the fields below are read */
  static chan_t chan;
  memset(&master, 0, sizeof master);
  memset(&chan, 0, sizeof chan);
  master.in_type = (enum filtertype)in_type;
  master.bins = master_bins;
  master.fdomain[0] = (float complex *)spectrum;
  chan.filter.out.master = &master;
  chan.filter.out.bins = slave_bins;
  chan.filter.out.next_jobnum = 1;      /* the function reads slot (next_jobnum - 1) % ND */
  Frontend.samprate = samprate;
  return estimate_noise(&chan, shift);
}
