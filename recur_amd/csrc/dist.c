/* dist.c -- the multi-GPU exchange step of the RNN core (gnu11 C).
 *
 * The reference sums the weight deltas of all the streams of a training set in one
 * process (recur-nn.c:724-739 over the sharing set up at recur-nn-init.c:232-241).  Here
 * the streams are sharded over one process per GPU; every rank computes the deltas of its
 * own streams into ih_delta||ho_delta and this file sums that ONE buffer over the ranks
 * with ONE all-reduce per generation (RCCL over xGMI), after which every rank applies the
 * identical update to its replica of the weights.
 *
 * RCCL is bound at rnn_amd_dist_init() time (dlopen of librccl.so.1 + dlsym), not at
 * link time: the single-GPU user, the host-only half of the API and the CPU test suite do
 * not map a 570 MB library they never call, and a process that already holds a copy of
 * RCCL (PyTorch bundles one) keeps exactly one.  Everything after the bind is a direct
 * call on the library's own stream.
 */
#include "rnn_host.h"
#include <dlfcn.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

/* the slice of rccl.h this file uses (ABI of RCCL 2.x: rccl/rccl.h:40-43, 187, 220, 260,
 * 339, 591, 611) */
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
enum { RAMD_NCCL_UINT8 = 1, RAMD_NCCL_FLOAT32 = 7, RAMD_NCCL_FLOAT64 = 8 };
enum { RAMD_NCCL_SUM = 0, RAMD_NCCL_MAX = 2 };

static struct {
  void *dl;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
  const char *(*GetErrorString)(ncclResult_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
} rccl;

static ncclComm_t g_comm = NULL;
static int g_rank = 0, g_world = 1;
static void *g_scratch = NULL; /* 256 device bytes for the small exchanges */

#define RCCL_OK(x)                                                                       \
  do {                                                                                   \
    ncclResult_t r_ = (x);                                                               \
    if (r_ != 0) {                                                                       \
      fprintf(stderr, "librecur_amd: RCCL error \"%s\" at %s:%d\n",                       \
              rccl.GetErrorString ? rccl.GetErrorString(r_) : "?", __FILE__, __LINE__); \
      abort();                                                                           \
    }                                                                                    \
  } while (0)
#define HIP_OK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "librecur_amd: HIP error \"%s\" at %s:%d\n", hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                       \
      abort();                                                                           \
    }                                                                                    \
  } while (0)

static int bind_rccl(void) {
  if (rccl.dl) {
    return 0;
  }
  const char *names[] = {getenv("RECUR_AMD_RCCL"), "librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so.1"};
  for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !rccl.dl; i++) {
    if (names[i] && *names[i]) {
      rccl.dl = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    }
  }
  if (!rccl.dl) {
    fprintf(stderr, "librecur_amd: cannot load librccl.so.1 (%s)\n", dlerror());
    return -1;
  }
#define SYM(field, name)                                                       \
  do {                                                                         \
    *(void **)&rccl.field = dlsym(rccl.dl, name);                              \
    if (!rccl.field) {                                                         \
      fprintf(stderr, "librecur_amd: librccl has no symbol %s\n", name);       \
      return -1;                                                               \
    }                                                                          \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(CommUserRank, "ncclCommUserRank");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Broadcast, "ncclBroadcast");
#undef SYM
  return 0;
}

int rnn_amd_dist_get_id(void *id) {
  if (bind_rccl()) {
    return -1;
  }
  ncclUniqueId u;
  RCCL_OK(rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

struct init_job {
  ncclUniqueId id;
  int rank, world, device;
  ncclComm_t comm;
  ncclResult_t result;
};
static void *init_thread(void *p) {
  struct init_job *j = p;
  if (hipSetDevice(j->device) != hipSuccess) { /* (a new thread starts on device 0) */
    j->result = 1;
    return NULL;
  }
  j->result = rccl.CommInitRank(&j->comm, j->world, j->id, j->rank);
  return NULL;
}

int rnn_amd_dist_init(int rank, int world, const void *id) {
  if (g_comm) {
    fprintf(stderr, "librecur_amd: rnn_amd_dist_init called twice\n");
    return -1;
  }
  if (world < 1 || rank < 0 || rank >= world || !id) {
    fprintf(stderr, "librecur_amd: rnn_amd_dist_init(rank %d, world %d)\n", rank, world);
    return -1;
  }
  ramd_require_device("rnn_amd_dist_init");
  ramd_note_side_stream(); /* RCCL's own queues and kernels run beside the library's from here on */
  if (bind_rccl()) {
    return -1;
  }
  /* ncclCommInitRank blocks until EVERY rank has arrived; one that never does (a rank that died, a
   * wrong id, ranks that cannot reach each other) would leave this process waiting for ever.  The call
   * runs on a helper thread and is given RECUR_AMD_RCCL_INIT_TIMEOUT seconds (default 180). */
  struct init_job job;
  memset(&job, 0, sizeof(job));
  memcpy(&job.id, id, sizeof(job.id));
  job.rank = rank;
  job.world = world;
  HIP_OK(hipGetDevice(&job.device));
  const char *te = getenv("RECUR_AMD_RCCL_INIT_TIMEOUT");
  long limit = 180;
  if (te) {
    limit = atol(te);
    if (limit <= 0) {
      fprintf(stderr, "librecur_amd: RECUR_AMD_RCCL_INIT_TIMEOUT=%s is not a number of seconds; using 180\n", te);
      limit = 180;
    }
  }
  pthread_t th;
  if (pthread_create(&th, NULL, init_thread, &job) != 0) {
    fprintf(stderr, "librecur_amd: rnn_amd_dist_init cannot start its helper thread\n");
    return -1;
  }
  struct timespec until;
  clock_gettime(CLOCK_REALTIME, &until);
  until.tv_sec += limit;
  if (pthread_timedjoin_np(th, NULL, &until) != 0) {
    fprintf(stderr, "librecur_amd: rank %d of %d has waited %ld s in ncclCommInitRank: not every rank joined the "
                    "group (is each of the %d ranks running, with rank 0's id, on a GPU of its own, "
                    "HSA_ENABLE_IPC_MODE_LEGACY=0 set?).  Giving up.\n", rank, world, limit, world);
    /* the helper thread sits inside RCCL and cannot be cancelled, and a process that has touched the GPU must
     * not re-exec: leave at once, without a core dump, with a status the launcher can read (documented in
     * include/recur_amd.h) */
    fflush(stderr);
    _exit(86);
  }
  if (job.result != 0) {
    fprintf(stderr, "librecur_amd: ncclCommInitRank failed: %s\n", rccl.GetErrorString(job.result));
    return -1;
  }
  g_comm = job.comm;
  /* what the communicator says, not what the caller said */
  RCCL_OK(rccl.CommCount(g_comm, &g_world));
  RCCL_OK(rccl.CommUserRank(g_comm, &g_rank));
  if (g_world != world || g_rank != rank) {
    fprintf(stderr, "librecur_amd: the RCCL group has %d ranks and this is rank %d (asked for rank %d of %d)\n",
            g_world, g_rank, rank, world);
    fflush(stderr);
    _exit(86); /* (a group that is not the one asked for: nothing this rank computes would be right) */
  }
  HIP_OK(hipMalloc(&g_scratch, 256));
  return 0;
}

void rnn_amd_dist_finalize(void) {
  if (g_comm) {
    rnn_amd_synchronize();
    RCCL_OK(rccl.CommDestroy(g_comm));
    g_comm = NULL;
    (void)hipFree(g_scratch);
    g_scratch = NULL;
  }
  g_rank = 0;
  g_world = 1;
}

int rnn_amd_dist_rank(void) { return g_rank; }
int rnn_amd_dist_world(void) { return g_world; }
int ramd_dist_active(void) { return g_comm != NULL; }

/* in-place sum over the ranks on a stream of the caller's choice (collectives of one communicator must
 * be issued in the same order on every rank: the callers see to that) */
void ramd_dist_all_reduce_on(void *device_buffer, size_t n_floats, void *stream) {
  if (!g_comm) {
    return; /* one process: the sum over ranks is the buffer itself */
  }
  RCCL_OK(rccl.AllReduce(device_buffer, device_buffer, n_floats, RAMD_NCCL_FLOAT32, RAMD_NCCL_SUM, g_comm,
                         (hipStream_t)stream));
}

void rnn_amd_dist_all_reduce(void *device_buffer, size_t n_floats) {
  ramd_dist_all_reduce_on(device_buffer, n_floats, rnn_amd_current_stream());
}

double rnn_amd_dist_max(double x) {
  if (!g_comm) {
    return x;
  }
  hipStream_t st = (hipStream_t)rnn_amd_current_stream();
  HIP_OK(hipMemcpyAsync(g_scratch, &x, sizeof(x), hipMemcpyHostToDevice, st));
  RCCL_OK(rccl.AllReduce(g_scratch, g_scratch, 1, RAMD_NCCL_FLOAT64, RAMD_NCCL_MAX, g_comm, st));
  HIP_OK(hipMemcpyAsync(&x, g_scratch, sizeof(x), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  return x;
}

void rnn_amd_dist_barrier(void) { (void)rnn_amd_dist_max(0.0); }

/* a few host bytes from `root` to every rank (generator states: see ramd_shared_rng) */
void ramd_dist_bcast(void *host, size_t bytes, int root) {
  if (!g_comm || g_world < 2) {
    return;
  }
  if (bytes > 256) {
    fprintf(stderr, "librecur_amd: ramd_dist_bcast of %zu bytes\n", bytes);
    abort();
  }
  hipStream_t st = (hipStream_t)rnn_amd_current_stream();
  if (g_rank == root) {
    HIP_OK(hipMemcpyAsync(g_scratch, host, bytes, hipMemcpyHostToDevice, st));
  }
  RCCL_OK(rccl.Broadcast(g_scratch, g_scratch, bytes, RAMD_NCCL_UINT8, root, g_comm, st));
  HIP_OK(hipMemcpyAsync(host, g_scratch, bytes, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
}

/* The prototype's generator is global stream 0's, so in a sharded set it lives on rank 0
 * (on the others nets[0]->rng is the generator of THEIR first stream).  Draws that every
 * replica has to make identically -- weight noise, perforation, the random damage of
 * rnn_condition_net -- therefore take rank 0's state: rank 0 draws from its own generator
 * (which advances, as in the reference), the others from a copy that is dropped.
 *
 * This is a COLLECTIVE (a 32-byte ncclBroadcast on the library's stream) and therefore only
 * taken for nets whose engine hosts a shard of a distributed training set (RamdEngine.sharded):
 * every rank of the group has to make the same call on its replica.  Any other net -- a
 * validation or confabulation clone family of its own, a side net that only one rank touches --
 * draws from its own generator as it does without a process group. */
rand_ctx *ramd_shared_rng(RecurNN *net, rand_ctx *tmp) {
  if (!g_comm || g_world < 2 || !ramd_engine_of(net)->sharded) {
    return &net->rng;
  }
  *tmp = net->rng;
  ramd_dist_bcast(tmp, sizeof(*tmp), 0);
  return g_rank == 0 ? &net->rng : tmp;
}
