/* filter_hip_mini.h -- small inline masters behind filter.h (included by filter_hip.c only).
 *
 * radiod's filter2 (src/radio.c:1503-1513,1572-1594; share/presets.conf:204,223,297): every channel that asks for it
 * owns a private COMPLEX master of N = round2(2*blocksize) points with ONE same-size COMPLEX slave, shift 0, optionally
 * ISB, and drives it inline from its own thread:
 *     write_cfilter(&filter2.in, first filter's output, olen)  ->  execute_filter_output(&filter2.out, 0)
 * A thousand such masters cannot be a thousand engines.  Here a COMPLEX master of N <= 8192 points is a "mini": it owns
 * nothing on the device but a response row in a pool shared by every mini of its geometry (chz_mini_*, include/chz_engine.h).
 *   execute_filter_input   only book-keeps (job number, where the block's N-sample window sits in the mirrored host ring)
 *                          and publishes the job at once -- there is nothing to wait for;
 *   execute_filter_output  queues the window for the pool's next launch.  The channel threads of one radiod block all
 *                          arrive here within microseconds of each other (they were released together), so the first one
 *                          becomes the batch leader and ONE kernel launch (one workgroup per instance) and ONE round trip
 *                          serve everybody who queued up meanwhile.
 * What a pooled instance cannot do -- a slave whose block size differs from its master's, or a REAL-output slave -- turns a
 * master that is still undecided (mini_wanted) into a full engine at that create_filter_output (round 6); on a master that
 * already serves a same-size slave it fails loudly. */
#define CTX_ENGINE 0x454e47
#define CTX_MINI   0x4d494e
#define CTX_SLAVE  0x534c56         /* struct sctx: a slave of an engine master */
#define CTX_MSLAVE 0x4d534c         /* struct msctx: a slave of a mini master */

struct mini_req {
  int inst, shift;
  unsigned char isb;
  const float *win;
  float *out;
  int rc;
  bool done;
  struct mini_req *next;
};

struct minipool {
  int L, M, cap, used;
  chz_mini *h;
  pthread_mutex_t lock;
  pthread_cond_t cv;
  struct mini_req *head, *tail;
  bool leader;
  struct minipool *next;
};

struct minictx {                    /* hangs off master->fwd_plan */
  int kind;                         /* CTX_MINI */
  int nslaves;                      /* slaves created on this master and not yet deleted */
  bool decided;                     /* a same-size COMPLEX slave has been created: this master IS a pooled inline master (see mini_wanted) */
  struct filter_in *master;
  const void *job_win[ND];          /* start of the N-sample window of the job in each slot */
};
struct msctx {                      /* hangs off slave->rev_plan */
  int kind;                         /* CTX_MSLAVE */
  struct minipool *pool;
  int inst;
};

static pthread_mutex_t Mini_registry_lock = PTHREAD_MUTEX_INITIALIZER;
static struct minipool *Mini_pools;

static bool smooth235(int n) { for (int p = 2; p <= 5; p++) while (p != 4 && n % p == 0) n /= p; return n == 1; }
/* A small COMPLEX master MAY be radiod's filter2 -- or the front end of a Funcube dongle (192 kHz: N = 4800) or an Airspy HF+ at its low rates.
   create_filter_input cannot tell, so such a master starts UNDECIDED (host state only): the first same-size COMPLEX create_filter_output makes it a
   pooled inline master (filter2 creates its slave right behind its master, src/radio.c:1583-1585); any other kind of slave, or a first block
   arriving while it has no slave at all (a front end streams before its channels exist), makes it a full engine -- in place: ring, pointers and
   job counter stay (round 6; rounds 2-5 refused every decimating slave of such a front end). */
static bool mini_wanted(int L, int M, enum filtertype in_type) {
  const char *e = XENV("KA9Q_HIP_MINI");
  if (e && e[0] == '0') return false;
  int const N = L + M - 1;
  return in_type == COMPLEX && N >= 8 && N <= 8192 && smooth235(N);
}
static bool is_mini_master(const struct filter_in *m) {
  if (!m) return false;
  const void *const ctx = __atomic_load_n((void *const *)(void *)&m->fwd_plan, __ATOMIC_ACQUIRE);      /* (swapped once, when an undecided master becomes an engine) */
  return ctx && *(const int *)ctx == CTX_MINI;
}

/* a pool of this geometry with a free instance (created on demand) */
static struct minipool *mini_pool_for(int L, int M) {
  pthread_mutex_lock(&Mini_registry_lock);
  struct minipool *p = Mini_pools;
  while (p && !(p->L == L && p->M == M && p->used < p->cap)) p = p->next;
  if (!p) {
    p = calloc(1, sizeof *p);
    if (p) {
      const char *e = XENV("KA9Q_HIP_MINI_POOL");
      const char *dev = getenv("KA9Q_HIP_DEVICE");
      p->L = L; p->M = M; p->cap = e && atoi(e) > 0 ? atoi(e) : 1024;
      if (chz_mini_create(&p->h, L, M, p->cap, dev ? atoi(dev) : 0) != 0) {
        fprintf(stderr, "filter_hip: mini-master pool L=%d M=%d: %s\n", L, M, chz_last_error());
        free(p); p = NULL;
      } else {
        pthread_mutex_init(&p->lock, NULL);
        pthread_cond_init(&p->cv, NULL);
        p->next = Mini_pools; Mini_pools = p;
      }
    }
  }
  if (p) p->used++;
  pthread_mutex_unlock(&Mini_registry_lock);
  return p;
}

static int mini_create_input(struct filter_in *master, int L, int M) {
  int const N = L + M - 1;
  struct minictx *c = calloc(1, sizeof *c);
  if (!c) return -1;
  c->kind = CTX_MINI; c->master = master;
  size_t const ring_bytes = page_round((size_t)ND * N * sizeof(float complex));      /* src/filter.c:237 */
  void *ring = ring_map(ring_bytes);
  void *fd[ND] = {NULL, NULL, NULL, NULL};
  bool ok = ring != NULL;
  for (int i = 0; ok && i < ND; i++) ok = (fd[i] = lmalloc(sizeof(float complex) * (size_t)N)) != NULL;
  if (!ok) { for (int i = 0; i < ND; i++) free(fd[i]); ring_unmap(&ring, ring_bytes); free(c); return -1; }
  master->points = N; master->bins = N; master->ilen = L; master->impulse_length = M;
  master->perform_inline = true;                                 /* a mini has no asynchronous half */
  for (int i = 0; i < ND; i++) {
    memset(fd[i], 0, sizeof(float complex) * (size_t)N);
    master->fdomain[i] = fd[i];                                   /* nobody reads a filter2 master's spectrum; kept allocated and zero */
    master->completed_jobs[i] = UINT_MAX;
  }
  if (!master->init) { pthread_mutex_init(&master->filter_mutex, NULL); pthread_cond_init(&master->filter_cond, NULL); master->init = true; }
  master->owner = pthread_self();
  master->in_type = COMPLEX;
  master->input_buffer_size = ring_bytes;
  master->input_buffer = ring;
  memset(ring, 0, ring_bytes);
  master->input_read_pointer.c = master->input_buffer;            /* src/filter.c:243-246 */
  master->input_write_pointer.c = master->input_read_pointer.c + (M - 1);
  master->input_read_pointer.r = NULL; master->input_write_pointer.r = NULL;
  master->wcnt = 0; master->next_jobnum = 0;
  master->fwd_plan = (fftwf_plan)(void *)c;
  return 0;
}

static void mini_free_input(struct filter_in *master) {           /* the ctx and buffers of a mini master */
  free((void *)master->fwd_plan); master->fwd_plan = NULL;
  ring_unmap(&master->input_buffer, master->input_buffer_size);
  for (int i = 0; i < ND; i++) { free(master->fdomain[i]); master->fdomain[i] = NULL; }
}

static int mini_create_output(struct filter_out *slave, struct filter_in *master, int len, enum filtertype out_type) {
  if (out_type == SPECTRUM) return 1;                              /* a block clock needs nothing: let the common path set it up */
  if (out_type != COMPLEX || len != master->ilen) {
    /* not what a pooled instance does (a decimating or REAL-output slave).  A master that has not run yet and has no slaves -- the usual
       order: create_filter_input, then its create_filter_output()s -- is simply re-made as a full engine by the caller (returns 2) */
    struct minictx const *mc = (struct minictx const *)(void const *)master->fwd_plan;
    if (!mc->decided && mc->nslaves == 0) return 2;
    fprintf(stderr, "create_filter_output: a %d-point inline master that already serves a same-size slave takes same-size COMPLEX slaves only (asked: olen %d, type %d)\n",
            master->points, len, (int)out_type);
    return -1;
  }
  struct minipool *p = mini_pool_for(master->ilen, master->impulse_length);
  if (!p) return -1;
  int inst = chz_mini_add(p->h);
  struct msctx *sc = calloc(1, sizeof *sc);
  float complex *buf = lmalloc(sizeof(float complex) * (size_t)master->points);
  float complex *fdom = lmalloc(sizeof(float complex) * (size_t)master->points);
  if (inst < 0 || !sc || !buf || !fdom) {
    if (inst >= 0) chz_mini_release(p->h, inst);
    pthread_mutex_lock(&Mini_registry_lock); p->used--; pthread_mutex_unlock(&Mini_registry_lock);
    free(sc); free(buf); free(fdom);
    return -1;
  }
  sc->kind = CTX_MSLAVE; sc->pool = p; sc->inst = inst;
  ((struct minictx *)(void *)master->fwd_plan)->nslaves++;
  __atomic_store_n(&((struct minictx *)(void *)master->fwd_plan)->decided, true, __ATOMIC_RELEASE);      /* (read without the mutex by execute_filter_input) */
  memset(buf, 0, sizeof(float complex) * (size_t)master->points);
  slave->bins = master->points;                                    /* src/filter.c:346 */
  slave->fdomain = fdom;
  slave->output_buffer.c = buf;
  slave->output.c = buf + slave->bins - len;                       /* src/filter.c:357 */
  slave->rev_plan = (fftwf_plan)(void *)sc;
  return 0;
}

static void mini_delete_output(struct filter_out *slave) {
  struct msctx *sc = (struct msctx *)(void *)slave->rev_plan;
  if (!sc) return;
  chz_mini_release(sc->pool->h, sc->inst);
  pthread_mutex_lock(&Mini_registry_lock); sc->pool->used--; pthread_mutex_unlock(&Mini_registry_lock);
  if (is_mini_master(slave->master)) ((struct minictx *)(void *)slave->master->fwd_plan)->nslaves--;       /* (a master deleted first is all zeros) */
  free(sc);
  slave->rev_plan = NULL;
}

static int mini_execute_input(struct filter_in *f) {
  struct minictx *c = (struct minictx *)(void *)f->fwd_plan;
  unsigned const job = __atomic_fetch_add(&f->next_jobnum, 1u, __ATOMIC_RELAXED);   /* src/filter.c:607; read lock-free by slaves being created */
  int const slot = (int)(job % ND);
  /* readers pick this up without a lock, possibly while a later lap overwrites it (as in the reference): tear-free accesses */
  __atomic_store_n(&f->samples_by_job[slot], f->sample_index, __ATOMIC_RELAXED);   /* src/filter.c:614-615 */
  f->sample_index += (uint64_t)f->ilen;
  c->job_win[slot] = f->input_read_pointer.c;                      /* N contiguous samples: the mirror sees to that */
  f->input_read_pointer.c += f->ilen;                              /* src/filter.c:626-636 */
  ring_wrap((void **)&f->input_read_pointer.c, f->input_buffer, f->input_buffer_size);
  pthread_mutex_lock(&f->filter_mutex);
  __atomic_store_n(&f->owner, pthread_self(), __ATOMIC_RELEASE);      /* read without the mutex by execute_filter_output */
  __atomic_store_n(&f->completed_jobs[slot], job, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&f->filter_cond);
  pthread_mutex_unlock(&f->filter_mutex);
  futex_wake_all(&f->completed_jobs[slot]);
  return 0;
}

static int mini_execute_output(struct filter_out *slave, int shift, int slot) {
  struct msctx *sc = (struct msctx *)(void *)slave->rev_plan;
  struct minictx *c = (struct minictx *)(void *)slave->master->fwd_plan;
  struct minipool *p = sc->pool;
  struct mini_req req = {.inst = sc->inst, .shift = shift, .isb = slave->isb ? 1 : 0,
                         .win = (const float *)c->job_win[slot], .out = (float *)slave->output.c};
  pthread_mutex_lock(&p->lock);
  if (p->tail) p->tail->next = &req; else p->head = &req;
  p->tail = &req;
  if (!p->leader) {
    p->leader = true;
    while (p->head) {
      struct mini_req *list = p->head;
      p->head = p->tail = NULL;
      pthread_mutex_unlock(&p->lock);
      int n = 0;
      for (struct mini_req *r = list; r; r = r->next) n++;
      int *inst = malloc(sizeof(int) * (size_t)n * 2);
      const float **win = malloc(sizeof(float *) * (size_t)n * 2);
      unsigned char *isb = malloc((size_t)n);
      int rc = (inst && win && isb) ? 0 : -1;
      if (rc == 0) {
        int *sh = inst + n; float **out = (float **)(win + n);
        int i = 0;
        for (struct mini_req *r = list; r; r = r->next, i++) { inst[i] = r->inst; sh[i] = r->shift; isb[i] = r->isb; win[i] = r->win; out[i] = r->out; }
        rc = chz_mini_execute(p->h, n, inst, win, sh, isb, out);
        if (rc != 0) fprintf(stderr, "execute_filter_output (inline master): %s\n", chz_last_error());
      }
      free(inst); free(win); free(isb);
      pthread_mutex_lock(&p->lock);
      for (struct mini_req *r = list; r;) { struct mini_req *nx = r->next; r->rc = rc; r->done = true; r = nx; }   /* r may vanish once done */
      pthread_cond_broadcast(&p->cv);
    }
    p->leader = false;
  } else {
    while (!req.done) pthread_cond_wait(&p->cv, &p->lock);
  }
  pthread_mutex_unlock(&p->lock);
  return req.rc == 0 ? 0 : -1;
}
