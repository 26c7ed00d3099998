// Does hipIpcGetMemHandle / hipIpcOpenMemHandle work on the kind of memory the tensor-parallel inboxes use
// (hipExtMallocWithFlags(hipDeviceMallocUncached)), between two processes, on this driver?  One GPU is enough to ask.
//   hipcc --offload-arch=gfx950 -o /tmp/ipc tools/ipc_probe.hip && /tmp/ipc
// The parent exports a handle, starts itself again as a child (before the child touches the GPU: a fresh process), the
// child maps the handle, fills the buffer with a kernel and exits, the parent checks what it sees.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>

__global__ void fill(unsigned* p, int n, unsigned v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void sum(const unsigned* p, int n, unsigned long long* out) { unsigned long long s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i]; atomicAdd(out, s); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = 1 << 16;
  if (argc == 3 && !strcmp(argv[1], "child")) {
    hipIpcMemHandle_t h;
    for (int i = 0; i < (int)sizeof(h); ++i) { unsigned b; sscanf(argv[2] + 2 * i, "%2x", &b); ((unsigned char*)&h)[i] = (unsigned char)b; }
    unsigned* p = nullptr;
    CK(hipIpcOpenMemHandle((void**)&p, h, hipIpcMemLazyEnablePeerAccess));
    hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, 0, p, n, 1000u);
    CK(hipDeviceSynchronize());
    CK(hipIpcCloseMemHandle(p));
    printf("child: mapped and filled\n");
    return 0;
  }
  for (int kind = 0; kind < 2; ++kind) {
    unsigned* p = nullptr;
    if (kind == 0) CK(hipExtMallocWithFlags((void**)&p, n * 4, hipDeviceMallocUncached)); else CK(hipMalloc(&p, n * 4));
    CK(hipMemset(p, 0, n * 4));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, p);
    printf("%s memory: hipIpcGetMemHandle -> %s\n", kind == 0 ? "uncached" : "ordinary", hipGetErrorString(e));
    if (e != hipSuccess) { (void)hipGetLastError(); continue; }
    char hex[2 * sizeof(h) + 1];
    for (int i = 0; i < (int)sizeof(h); ++i) sprintf(hex + 2 * i, "%02x", ((unsigned char*)&h)[i]);
    char cmd[1024];
    snprintf(cmd, sizeof(cmd), "%s child %s", argv[0], hex);
    const int rc = system(cmd);                                   // a fresh process, not a fork of this GPU-initialised one
    unsigned long long* d; CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(sum, dim3(1), dim3(256), 0, 0, p, n, d);
    unsigned long long s = 0; CK(hipMemcpy(&s, d, 8, hipMemcpyDeviceToHost));
    const unsigned long long want = (unsigned long long)n * 1000 + (unsigned long long)n * (n - 1) / 2;
    printf("%s memory: child exit %d, parent sees %s\n", kind == 0 ? "uncached" : "ordinary", WEXITSTATUS(rc), s == want ? "the child's data" : "something else");
  }
  return 0;
}
