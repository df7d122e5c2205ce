/* cgo_shape_check.c -- the C ABI driven the way go/xgmi/xgmi.go drives it, without Go.
 *
 * The image has no Go toolchain, so the cgo shim cannot be compiled here.  What cgo imposes on a C library can be
 * tested from C, though (mpi.go:121-125):
 *   - goroutines migrate between OS threads: EVERY call below is made from a freshly created pthread that has
 *     never selected a HIP device (and never will make another call);
 *   - out-parameters are stack variables of the calling thread (cgo passes &n, &dt, &req of the Go frame);
 *   - an empty Go slice arrives as (NULL, 0);
 *   - C must not keep a host pointer after the call returns: host buffers are overwritten right after each call;
 *   - concurrent Send / Receive from different threads with distinct {peer, tag} (helloworld.go:53-81 does that).
 * The call sequence follows xgmi.go method by method: Init, Rank, Size, Send / Receive of host slices (probe first,
 * as Receive does), SendNoWait / Wait, the collectives on host slices and on DeviceBuffers, IAllreduce finished from
 * another thread, Register / Deregister, the stream-ordered forms, Finalize.  One rank per process:
 *     xmpirun 2 tests/cgo_shape_check_bin
 * Exit status 0 = every call returned what the shim expects.  Compiled with gcc (plain C, like cgo's view of the header).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xmpi.h"

typedef void (*call_fn)(void*);
static void* trampoline(void* p) {
  void** a = (void**)p;
  ((call_fn)a[0])(a[1]);
  return NULL;
}
/* run fn(arg) on a brand-new OS thread and wait for it: what a goroutine scheduled onto a fresh M looks like to C */
static void on_new_thread(call_fn fn, void* arg) {
  pthread_t t;
  void* a[2] = {(void*)fn, arg};
  if (pthread_create(&t, NULL, trampoline, a) != 0) abort();
  pthread_join(t, NULL);
}

static xmpi_comm* g_comm;
static int g_rank, g_size, g_fail;
static char g_err[512]; /* xmpi_last_error() is per thread: the calling thread leaves its text here */
static void keep_error(int rc) {
  if (rc != XMPI_OK) snprintf(g_err, sizeof g_err, "%s; %s", xmpi_strerror(rc), xmpi_last_error());
}
#define EXPECT(cond, what)                                                                         \
  do {                                                                                             \
    if (!(cond)) {                                                                                 \
      fprintf(stderr, "rank %d: %s failed (%s:%d): %s\n", g_rank, what, __FILE__, __LINE__, g_err);             \
      g_err[0] = 0;                                                                                \
      g_fail++;                                                                                    \
    }                                                                                              \
  } while (0)

static int parse_rank(int argc, char** argv, int* size) { /* rank = index of -mpi-addr in the sorted -mpi-alladdr list */
  const char *addr = NULL, *all = NULL;
  for (int i = 1; i + 1 < argc; i++) {
    if (!strcmp(argv[i], "-mpi-addr")) addr = argv[i + 1];
    if (!strcmp(argv[i], "-mpi-alladdr")) all = argv[i + 1];
  }
  if (!addr || !all) {
    *size = 1;
    return 0;
  }
  char* copy = strdup(all);
  char* list[64];
  int n = 0;
  for (char* t = strtok(copy, ","); t && n < 64; t = strtok(NULL, ",")) list[n++] = t;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (strcmp(list[j], list[i]) < 0) {
        char* x = list[i];
        list[i] = list[j];
        list[j] = x;
      }
  int rank = -1;
  for (int i = 0; i < n; i++)
    if (!strcmp(list[i], addr)) rank = i;
  *size = n;
  return rank;
}

struct init_args { int rank, size; const char* key; };
static void do_init(void* p) {
  struct init_args* a = (struct init_args*)p;
  const char* dev = getenv("XMPI_DEVICE");
  EXPECT(xmpi_rank(NULL) == -1 && xmpi_size(NULL) == 0, "Rank / Size before Init (mpi.go:110-111)");
  EXPECT(xmpi_init(a->rank, a->size, dev ? atoi(dev) : -1, a->key, &g_comm) == XMPI_OK, "xmpi_init");
}
static void do_rank_size(void* p) {
  (void)p;
  g_rank = xmpi_rank(g_comm);
  g_size = xmpi_size(g_comm);
}

/* ---- Send / Receive of host slices ---- */
struct p2p_args { void* buf; size_t n; xmpi_dtype dt; int peer, tag; int rc; size_t got; };
static void do_send(void* p) {
  struct p2p_args* a = (struct p2p_args*)p;
  a->rc = xmpi_send(g_comm, a->buf, a->n, a->dt, a->peer, a->tag);
}
static void do_send_nowait(void* p) {
  struct p2p_args* a = (struct p2p_args*)p;
  a->rc = xmpi_send_nowait(g_comm, a->buf, a->n, a->dt, a->peer, a->tag);
}
static void do_wait(void* p) {
  struct p2p_args* a = (struct p2p_args*)p;
  a->rc = xmpi_wait(g_comm, a->peer, a->tag);
}
static void do_recv_like_go(void* p) { /* Receive(*[]T): probe, size the slice, receive in place */
  struct p2p_args* a = (struct p2p_args*)p;
  size_t n = 0;           /* stack out-parameters of THIS thread */
  xmpi_dtype dt = XMPI_U8;
  a->rc = xmpi_probe(g_comm, a->peer, a->tag, &n, &dt);
  if (a->rc != XMPI_OK) return;
  if (dt != a->dt) {
    a->rc = XMPI_ERR_ARG;
    return;
  }
  a->buf = n ? malloc(n * xmpi_dtype_size(dt)) : NULL; /* empty slice: nil pointer */
  size_t got = 0;
  a->rc = xmpi_recv(g_comm, a->buf, n, dt, a->peer, a->tag, &got);
  a->got = got;
  a->n = n;
}
static void* send_thread(void* p) {
  do_send(p);
  return NULL;
}
static void* recv_thread(void* p) {
  do_recv_like_go(p);
  return NULL;
}

/* ---- collectives ---- */
struct coll_args { int which; const void* s; void* r; size_t n; xmpi_dtype dt; int op, root; void* stream; int rc; xmpi_request* req; };
static void do_coll(void* p) {
  struct coll_args* a = (struct coll_args*)p;
  switch (a->which) {
    case 0: a->rc = xmpi_allreduce(g_comm, a->s, a->r, a->n, a->dt, (xmpi_op)a->op, XMPI_ALGO_AUTO); break;
    case 1: a->rc = xmpi_allgather(g_comm, a->s, a->r, a->n, a->dt, XMPI_ALGO_AUTO); break;
    case 2: a->rc = xmpi_bcast(g_comm, a->r, a->n, a->dt, a->root, XMPI_ALGO_AUTO); break;
    case 3: a->rc = xmpi_reduce(g_comm, a->s, a->r, a->n, a->dt, (xmpi_op)a->op, a->root, XMPI_ALGO_AUTO); break;
    case 4: a->rc = xmpi_barrier(g_comm); break;
    case 5: {
      xmpi_request* req = NULL; /* &req of the caller's frame */
      a->rc = xmpi_iallreduce(g_comm, a->s, a->r, a->n, a->dt, (xmpi_op)a->op, XMPI_ALGO_AUTO, &req);
      a->req = req;
      break;
    }
    case 6: a->rc = xmpi_request_wait(a->req); break; /* the goroutine that completes the channel: another thread */
    case 7: a->rc = xmpi_allreduce_on_stream(g_comm, a->s, a->r, a->n, a->dt, (xmpi_op)a->op, a->stream); break;
    case 8: a->rc = xmpi_stream_sync(g_comm, a->stream); break;
    case 9: a->rc = xmpi_allgather_on_stream(g_comm, a->s, a->r, a->n, a->dt, a->stream); break;
    case 10: a->rc = xmpi_bcast_on_stream(g_comm, a->r, a->n, a->dt, a->root, a->stream); break;
    case 11: a->rc = xmpi_reduce_on_stream(g_comm, a->s, a->r, a->n, a->dt, (xmpi_op)a->op, a->root, a->stream); break;
    case 12: a->rc = xmpi_graph_begin(g_comm, a->stream); break;
    case 13: {
      void* g = NULL; /* out-parameter of the caller's frame */
      a->rc = xmpi_graph_end(g_comm, a->stream, &g);
      a->r = g;
      break;
    }
    case 14: a->rc = xmpi_graph_launch(g_comm, a->r, a->stream); break;
    case 15: a->rc = xmpi_graph_destroy(g_comm, a->r); break;
  }
  keep_error(a->rc);
}

struct mem_args { void* dst; const void* src; size_t bytes; void* out; int rc; };
static void do_malloc(void* p) {
  struct mem_args* a = (struct mem_args*)p;
  a->out = xmpi_malloc(g_comm, a->bytes);
}
static void do_free(void* p) {
  struct mem_args* a = (struct mem_args*)p;
  a->rc = xmpi_free(g_comm, a->dst);
}
static void do_memcpy(void* p) {
  struct mem_args* a = (struct mem_args*)p;
  a->rc = xmpi_memcpy(g_comm, a->dst, a->src, a->bytes);
}
static void do_stream_create(void* p) {
  struct mem_args* a = (struct mem_args*)p;
  a->out = xmpi_stream_create(g_comm);
}
static void do_stream_destroy(void* p) {
  struct mem_args* a = (struct mem_args*)p;
  a->rc = xmpi_stream_destroy(g_comm, a->dst);
}
static void do_finalize(void* p) {
  (void)p;
  EXPECT(xmpi_finalize(g_comm) == XMPI_OK, "xmpi_finalize");
}

static int coll(int which, const void* s, void* r, size_t n, xmpi_dtype dt, int op, int root, void* stream) {
  struct coll_args a = {which, s, r, n, dt, op, root, stream, 0, NULL};
  on_new_thread(do_coll, &a);
  return a.rc;
}

int main(int argc, char** argv) {
  struct init_args ia;
  ia.rank = parse_rank(argc, argv, &ia.size);
  char key[96];
  snprintf(key, sizeof key, "%scgo", getenv("XMPI_JOB") ? getenv("XMPI_JOB") : "solo");
  ia.key = key;
  if (ia.rank < 0) return 2;
  on_new_thread(do_init, &ia);
  if (!g_comm) return 1;
  on_new_thread(do_rank_size, NULL);
  EXPECT(g_rank == ia.rank && g_size == ia.size, "Rank / Size");
  const int me = g_rank, N = g_size;

  /* 1. helloworld's idiom: concurrent Send + Receive with every rank (self included), one thread each, tag 0;
   *    float64 payloads of different lengths, rank r sends r+3 elements; rank 0's peers also get an EMPTY slice on tag 1 */
  {
    pthread_t ts[64], tr[64];
    struct p2p_args sa[32], ra[32];
    double* out[32];
    for (int p = 0; p < N; p++) {
      const size_t n = (size_t)me + 3;
      out[p] = (double*)malloc(n * 8);
      for (size_t i = 0; i < n; i++) out[p][i] = 1000.0 * me + p + 0.5 * (double)i;
      sa[p] = (struct p2p_args){out[p], n, XMPI_F64, p, 0, 0, 0};
      ra[p] = (struct p2p_args){NULL, 0, XMPI_F64, p, 0, 0, 0};
      pthread_create(&ts[p], NULL, send_thread, &sa[p]);
      pthread_create(&tr[p], NULL, recv_thread, &ra[p]);
    }
    for (int p = 0; p < N; p++) {
      pthread_join(ts[p], NULL);
      pthread_join(tr[p], NULL);
      EXPECT(sa[p].rc == XMPI_OK, "Send (concurrent)");
      EXPECT(ra[p].rc == XMPI_OK && ra[p].n == (size_t)p + 3 && ra[p].got == ra[p].n, "Receive (concurrent, sized by probe)");
      for (size_t i = 0; ra[p].rc == XMPI_OK && i < ra[p].n; i++)
        EXPECT(((double*)ra[p].buf)[i] == 1000.0 * p + me + 0.5 * (double)i, "payload of Receive");
      memset(out[p], 0xEE, ((size_t)me + 3) * 8); /* Go may reuse / collect the slice right after Send returned */
      free(out[p]);
      free(ra[p].buf);
    }
    if (N > 1) { /* empty []byte: (NULL, 0) both ways */
      const int peer = me ^ 1;
      if (peer < N) {
        struct p2p_args s = {NULL, 0, XMPI_U8, peer, 1, 0, 0}, r = {NULL, 0, XMPI_U8, peer, 1, 0, 0};
        pthread_t a, b;
        pthread_create(&a, NULL, send_thread, &s);
        pthread_create(&b, NULL, recv_thread, &r);
        pthread_join(a, NULL);
        pthread_join(b, NULL);
        EXPECT(s.rc == XMPI_OK && r.rc == XMPI_OK && r.n == 0 && r.buf == NULL, "empty slice round trip");
      }
    }
  }
  /* 2. SendNoWait / Wait (mpi.go:132-152) with the neighbour; a duplicate {dest, tag} must be refused, not crash */
  if (N > 1 && (me ^ 1) < N) {
    const int peer = me ^ 1;
    int32_t v[5] = {me, 1, 2, 3, 4};
    struct p2p_args s = {v, 5, XMPI_I32, peer, 7, 0, 0}, dup = {v, 5, XMPI_I32, peer, 7, 0, 0}, r = {NULL, 0, XMPI_I32, peer, 7, 0, 0};
    on_new_thread(do_send_nowait, &s);
    EXPECT(s.rc == XMPI_OK, "SendNoWait");
    on_new_thread(do_send_nowait, &dup);
    EXPECT(dup.rc == XMPI_ERR_TAG_EXISTS, "duplicate {dest, tag} -> mpi.TagExists");
    memset(v, 0x7F, sizeof v); /* the payload has left the buffer */
    on_new_thread(do_recv_like_go, &r);
    EXPECT(r.rc == XMPI_OK && r.n == 5 && ((int32_t*)r.buf)[0] == peer && ((int32_t*)r.buf)[4] == 4, "Receive of SendNoWait");
    free(r.buf);
    on_new_thread(do_wait, &s);
    EXPECT(s.rc == XMPI_OK, "Wait");
  }
  /* 3. collectives on plain host slices (what the examples of the reference would pass) */
  {
    float x[1000], y[1000];
    for (int i = 0; i < 1000; i++) x[i] = (float)(me + 1) + (float)(i % 7);
    EXPECT(coll(0, x, y, 1000, XMPI_F32, XMPI_SUM, 0, NULL) == XMPI_OK, "Allreduce(host)");
    for (int i = 0; i < 1000; i++) EXPECT(y[i] == (float)(N * (N + 1) / 2) + (float)N * (float)(i % 7), "Allreduce(host) value");
    int64_t mine[2] = {me, 42}, all[64];
    EXPECT(coll(1, mine, all, 2, XMPI_I64, 0, 0, NULL) == XMPI_OK, "Allgather(host)");
    for (int r = 0; r < N; r++) EXPECT(all[2 * r] == r && all[2 * r + 1] == 42, "Allgather(host) value");
    double tok = me == N - 1 ? 2.5 : 0.0;
    EXPECT(coll(2, NULL, &tok, 1, XMPI_F64, 0, N - 1, NULL) == XMPI_OK && tok == 2.5, "Bcast(host)");
    int64_t bit = (int64_t)1 << me, mask = 0;
    EXPECT(coll(3, &bit, me == 0 ? &mask : NULL, 1, XMPI_I64, XMPI_SUM, 0, NULL) == XMPI_OK, "Reduce(host)");
    EXPECT(me != 0 || mask == ((int64_t)1 << N) - 1, "Reduce(host) value");
    EXPECT(coll(0, NULL, NULL, 0, XMPI_F32, XMPI_SUM, 0, NULL) == XMPI_OK, "Allreduce of empty slices (nil, 0)");
    EXPECT(coll(4, NULL, NULL, 0, XMPI_U8, 0, 0, NULL) == XMPI_OK, "Barrier");
  }
  /* 4. DeviceBuffers: Malloc / Upload / collectives / IAllreduce completed by another thread / streams / Download / Free */
  {
    const size_t n = 100003;
    struct mem_args ms = {NULL, NULL, n * 4, NULL, 0}, mr = ms, mg = {NULL, NULL, n * 4 * (size_t)N, NULL, 0};
    on_new_thread(do_malloc, &ms);
    on_new_thread(do_malloc, &mr);
    on_new_thread(do_malloc, &mg);
    EXPECT(ms.out && mr.out && mg.out, "Malloc");
    float* host = (float*)malloc(n * 4 * (size_t)N);
    for (size_t i = 0; i < n; i++) host[i] = (float)(me + 1) + (float)(i % 5);
    struct mem_args up = {ms.out, host, n * 4, NULL, 0};
    on_new_thread(do_memcpy, &up);
    EXPECT(up.rc == XMPI_OK, "Upload");
    memset(host, 0, n * 4);
    EXPECT(coll(0, ms.out, mr.out, n, XMPI_F32, XMPI_SUM, 0, NULL) == XMPI_OK, "Allreduce(DeviceBuffer)");
    struct coll_args ia2 = {5, ms.out, mr.out, n, XMPI_F32, XMPI_MAX, 0, NULL, 0, NULL};
    on_new_thread(do_coll, &ia2);
    EXPECT(ia2.rc == XMPI_OK && ia2.req, "IAllreduce");
    struct coll_args wa = {6, NULL, NULL, 0, XMPI_F32, 0, 0, NULL, 0, ia2.req};
    on_new_thread(do_coll, &wa);
    EXPECT(wa.rc == XMPI_OK, "request wait from another thread");
    struct mem_args dn = {host, mr.out, n * 4, NULL, 0};
    on_new_thread(do_memcpy, &dn);
    for (size_t i = 0; i < n; i += 997) EXPECT(host[i] == (float)N + (float)(i % 5), "IAllreduce(MAX) value");
    /* stream-ordered forms, enqueued from one thread, waited for from another */
    struct mem_args st = {NULL, NULL, 0, NULL, 0};
    on_new_thread(do_stream_create, &st);
    EXPECT(st.out != NULL, "Stream()");
    EXPECT(coll(7, ms.out, mr.out, n, XMPI_F32, XMPI_SUM, 0, st.out) == XMPI_OK, "AllreduceOnStream");
    EXPECT(coll(9, mr.out, mg.out, n, XMPI_F32, 0, 0, st.out) == XMPI_OK, "AllgatherOnStream");
    EXPECT(coll(10, NULL, ms.out, n, XMPI_F32, 0, 0, st.out) == XMPI_OK, "BcastOnStream");
    EXPECT(coll(11, mg.out, me == N - 1 ? mg.out : NULL, n * (size_t)N, XMPI_F32, XMPI_MIN, N - 1, st.out) == XMPI_OK, "ReduceOnStream");
    EXPECT(coll(8, NULL, NULL, 0, XMPI_F32, 0, 0, st.out) == XMPI_OK, "StreamSync");
    struct mem_args dg = {host, mg.out, n * 4 * (size_t)N, NULL, 0};
    on_new_thread(do_memcpy, &dg);
    for (int r = 0, said = 0; r < N; r++)
      for (size_t i = 0; i < n; i += 1009) {
        const float want = (float)(N * (N + 1) / 2) + (float)N * (float)(i % 5);
        if (host[(size_t)r * n + i] != want && said++ < 4)
          fprintf(stderr, "rank %d: gathered block %d element %zu = %g, want %g\n", me, r, i, host[(size_t)r * n + i], want);
        EXPECT(host[(size_t)r * n + i] == want || said > 4, "AllgatherOnStream / ReduceOnStream value");
      }
    if (N > 1) { /* GraphBegin ... GraphEnd, GraphLaunch x3: begin, capture, end and every replay from different threads */
      EXPECT(coll(7, ms.out, mr.out, n, XMPI_F32, XMPI_MAX, 0, st.out) == XMPI_OK, "AllreduceOnStream (warm)");
      EXPECT(coll(8, NULL, NULL, 0, XMPI_F32, 0, 0, st.out) == XMPI_OK, "StreamSync");
      EXPECT(coll(12, NULL, NULL, 0, XMPI_F32, 0, 0, st.out) == XMPI_OK, "GraphBegin");
      EXPECT(coll(7, ms.out, mr.out, n, XMPI_F32, XMPI_MAX, 0, st.out) == XMPI_OK, "AllreduceOnStream (captured)");
      struct coll_args ge = {13, NULL, NULL, 0, XMPI_F32, 0, 0, st.out, 0, NULL};
      on_new_thread(do_coll, &ge);
      EXPECT(ge.rc == XMPI_OK && ge.r != NULL, "GraphEnd");
      for (int k = 0; k < 3; k++) EXPECT(coll(14, NULL, ge.r, 0, XMPI_F32, 0, 0, st.out) == XMPI_OK, "GraphLaunch");
      EXPECT(coll(8, NULL, NULL, 0, XMPI_F32, 0, 0, st.out) == XMPI_OK, "StreamSync after replays");
      EXPECT(coll(15, NULL, ge.r, 0, XMPI_F32, 0, 0, NULL) == XMPI_OK, "GraphDestroy");
    }
    struct mem_args sd = {st.out, NULL, 0, NULL, 0};
    on_new_thread(do_stream_destroy, &sd);
    struct mem_args f1 = {ms.out, NULL, 0, NULL, 0}, f2 = {mr.out, NULL, 0, NULL, 0}, f3 = {mg.out, NULL, 0, NULL, 0};
    on_new_thread(do_free, &f1);
    on_new_thread(do_free, &f2);
    on_new_thread(do_free, &f3);
    EXPECT(f1.rc == XMPI_OK && f2.rc == XMPI_OK && f3.rc == XMPI_OK, "Free");
    free(host);
  }
  EXPECT(coll(4, NULL, NULL, 0, XMPI_U8, 0, 0, NULL) == XMPI_OK, "Barrier");
  on_new_thread(do_finalize, NULL);
  if (me == 0 && !g_fail) printf("cgo shape check: %d ranks, every call from a fresh OS thread: ok\n", N);
  return g_fail ? 1 : 0;
}
