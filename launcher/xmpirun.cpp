// xmpirun -- one-process-per-GPU launcher; replaces mpirun/gompirun/gompirun.go:28-93.
//
//   xmpirun N prog [args...]
//
// Like the reference's gompirun it starts N copies of `prog` on this machine, gives copy i the
// address ":6000+i" and every copy the full list (-mpi-addr / -mpi-alladdr: the launcher <->
// program contract of flags.go:44-50; the rank is the index in the sorted list,
// network.go:94-109), shares stdin/stdout/stderr and waits for all of them.  Differences, all on
// purpose:
//   * rank i is pinned to GPU i % G (XMPI_DEVICE; G = $XMPI_NGPUS, or the GPUs of the KFD topology that ROCR_ / HIP_VISIBLE_DEVICES
//     leave visible) -- "one rank owns one MI355X"; `xmpirun: N ranks on G GPUs` says so once on stderr;
//   * when ranks share a GPU (N > G) each copy is limited to 2 hardware queues (GPU_MAX_HW_QUEUES, unless set);
//   * every copy gets the same fresh job id (XMPI_JOB) so two jobs never meet in one control block;
//   * the exit status is the worst child status (the reference drops it: gompirun.go:89);
//   * SIGTERM / SIGINT / SIGHUP are passed on to the ranks.
#include <dirent.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

// a launcher that is told to stop takes its ranks with it (a rank left behind keeps its GPU queue and memory)
static pid_t g_kids[256];
static int g_nkids = 0;
static void forward_signal(int sig) {
  for (int i = 0; i < g_nkids; i++)
    if (g_kids[i] > 0) kill(g_kids[i], sig == SIGINT ? SIGINT : SIGTERM);
}

// How many entries of a *_VISIBLE_DEVICES list name a device the level below shows (`have` of them): indices 0 .. have-1, or
// UUIDs ("GPU-...": taken at their word); the list ends at the first entry that names none (the runtimes' rule).  Unset: all.
static int visible(const char* name, int have) {
  const char* e = getenv(name);
  if (!e) return have;
  int n = 0;
  std::string list = e;
  size_t pos = 0;
  while (pos <= list.size()) {
    size_t comma = list.find(',', pos);
    if (comma == std::string::npos) comma = list.size();
    std::string tok = list.substr(pos, comma - pos);
    while (!tok.empty() && tok.front() == ' ') tok.erase(0, 1);
    while (!tok.empty() && tok.back() == ' ') tok.pop_back();
    pos = comma + 1;
    if (tok.empty()) break;
    if (!strncmp(tok.c_str(), "GPU-", 4)) {
      n++;
      continue;
    }
    char* end = nullptr;
    const long idx = strtol(tok.c_str(), &end, 10);
    if (!end || *end || idx < 0 || idx >= have) break;
    n++;
  }
  return n;
}

// The GPUs this job's processes will see.  $XMPI_NGPUS if given; otherwise the KFD topology -- the nodes with SIMDs are GPUs (a CPU
// node has simd_count 0; a render node of another vendor's GPU has no KFD node at all) -- narrowed by ROCR_VISIBLE_DEVICES (which
// GPUs the ROCm runtime shows) and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (which of THOSE the HIP runtime shows): with 4 of
// 8 GPUs visible rank 5 must get XMPI_DEVICE = 1, not 5 ("device 5 does not exist").  No KFD (a container without /sys/class/kfd):
// the /dev/dri render nodes, as before.  (gompirun.go:57-93 starts N copies on one machine and knows nothing of devices.)
static int count_gpus(std::string* how) {
  if (const char* e = getenv("XMPI_NGPUS")) {
    *how = "XMPI_NGPUS";
    return atoi(e) > 0 ? atoi(e) : 1;
  }
  int n = 0;
  const char* root = getenv("XMPI_KFD_TOPOLOGY");  // (tests: a topology directory of their own)
  const std::string nodes = std::string(root ? root : "/sys/class/kfd/kfd/topology") + "/nodes";
  if (DIR* d = opendir(nodes.c_str())) {
    while (dirent* ent = readdir(d)) {
      if (ent->d_name[0] == '.') continue;
      FILE* f = fopen((nodes + "/" + ent->d_name + "/properties").c_str(), "r");
      if (!f) continue;
      char key[64];
      unsigned long long val = 0;
      while (fscanf(f, "%63s %llu", key, &val) == 2)
        if (!strcmp(key, "simd_count") && val > 0) {
          n++;
          break;
        }
      fclose(f);
    }
    closedir(d);
    *how = "KFD topology";
  }
  if (n == 0) {
    if (DIR* d = opendir("/dev/dri")) {
      while (dirent* ent = readdir(d))
        if (!strncmp(ent->d_name, "renderD", 7)) n++;
      closedir(d);
    }
    *how = "/dev/dri render nodes";
  }
  if (n == 0) {
    *how = "no GPU found: assuming one";
    return 1;
  }
  const int all = n;
  n = visible("ROCR_VISIBLE_DEVICES", n);
  n = visible(getenv("HIP_VISIBLE_DEVICES") ? "HIP_VISIBLE_DEVICES" : "CUDA_VISIBLE_DEVICES", n);
  if (n != all) *how += ", " + std::to_string(n) + " of " + std::to_string(all) + " visible (ROCR_ / HIP_VISIBLE_DEVICES)";
  return n > 0 ? n : 1;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "less than two arguments, must have at least executable and number of nodes\nusage: %s N prog [args...]\n",
            argv[0]);
    return 2;
  }
  const int n = atoi(argv[1]);
  if (n < 1) {
    fprintf(stderr, "number of nodes must be positive\n");
    return 2;
  }
  const int base = getenv("XMPI_BASEPORT") ? atoi(getenv("XMPI_BASEPORT")) : 6000;  // gompirun.go:46
  std::vector<std::string> ports;
  std::string all;
  for (int i = 0; i < n; i++) {
    ports.push_back(":" + std::to_string(base + i));
    all += (i ? "," : "") + ports.back();
  }
  char job[64];
  snprintf(job, sizeof job, "x%lx-%x-", (long)time(nullptr), (unsigned)getpid());
  std::string how;
  const int gpus = count_gpus(&how);
  fprintf(stderr, "xmpirun: %d ranks on %d GPUs (%s)%s\n", n, gpus, how.c_str(), n > gpus ? ": several ranks per GPU" : "");
  if (const char* d = getenv("XMPI_DEVICE"))  // (set for every rank by whoever started the launcher: one device for all of them)
    if (atoi(d) < 0 || atoi(d) >= gpus) {
      fprintf(stderr, "xmpirun: XMPI_DEVICE=%s, but the ranks will see %d GPU%s (devices 0 .. %d): unset it, or name one of those\n", d, gpus,
              gpus == 1 ? "" : "s", gpus - 1);
      return 2;
    }
  std::vector<pid_t> kids;
  for (int i = 0; i < n; i++) {
    pid_t pid = fork();
    if (pid < 0) {
      perror("fork");
      return 1;
    }
    if (pid == 0) {
      // ports sort lexicographically like their numbers as long as they have equal width
      setenv("XMPI_JOB", job, 1);
      // for programs on the bare C ABI (the Go / C++ front ends derive the rank from the -mpi-* flags: network.go:94-109;
      // the ports below sort like their numbers, so that rank is i as well)
      setenv("XMPI_RANK", std::to_string(i).c_str(), 1);
      setenv("XMPI_SIZE", std::to_string(n).c_str(), 1);
      setenv("XMPI_DEVICE", std::to_string(i % gpus).c_str(), 0);
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
      // ranks sharing a GPU share its hardware queues too: a process may hold 4 by default, and beyond a few
      // dozen in total the GPU's scheduler time-slices them (milliseconds per collective instead of microseconds)
      if (n > gpus) setenv("GPU_MAX_HW_QUEUES", "2", 0);
      std::vector<char*> av;
      av.push_back(argv[2]);
      for (int k = 3; k < argc; k++) av.push_back(argv[k]);
      av.push_back((char*)"-mpi-addr");
      av.push_back((char*)ports[(size_t)i].c_str());
      av.push_back((char*)"-mpi-alladdr");
      av.push_back((char*)all.c_str());
      av.push_back(nullptr);
      execvp(argv[2], av.data());
      perror(argv[2]);
      _exit(127);
    }
    kids.push_back(pid);
    if (g_nkids < 256) g_kids[g_nkids++] = pid;
  }
  signal(SIGTERM, forward_signal);
  signal(SIGINT, forward_signal);
  signal(SIGHUP, forward_signal);
  int worst = 0;
  for (pid_t k : kids) {
    int st = 0;
    while (waitpid(k, &st, 0) < 0 && errno == EINTR) {
    }
    int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (code > worst) worst = code;
  }
  return worst;
}
