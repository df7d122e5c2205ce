// xmpirun -- one-process-per-GPU launcher; replaces mpirun/gompirun/gompirun.go:28-93.
//
//   xmpirun N prog [args...]
//
// Like the reference's gompirun it starts N copies of `prog` on this machine, gives copy i the
// address ":6000+i" and every copy the full list (-mpi-addr / -mpi-alladdr: the launcher <->
// program contract of flags.go:44-50; the rank is the index in the sorted list,
// network.go:94-109), shares stdin/stdout/stderr and waits for all of them.  Differences, all on
// purpose:
//   * rank i is pinned to GPU i % G (XMPI_DEVICE; G = $XMPI_NGPUS or the number of /dev/dri
//     render nodes) -- "one rank owns one MI355X";
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

static int count_gpus() {
  if (const char* e = getenv("XMPI_NGPUS")) return atoi(e) > 0 ? atoi(e) : 1;
  int n = 0;
  if (DIR* d = opendir("/dev/dri")) {
    while (dirent* ent = readdir(d))
      if (!strncmp(ent->d_name, "renderD", 7)) n++;
    closedir(d);
  }
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
  const int gpus = count_gpus();
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
