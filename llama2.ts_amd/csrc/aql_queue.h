// aql_queue.h -- the decode step as hand-written AQL packets on a user-mode HSA queue (aql_queue.cpp).
//
// A token of the greedy loop (llama2.ts:465-508 at -t 0) is 4L + 2 ... 5L + 2 DEPENDENT kernel launches, and at the small models
// the boundary between two of them is 40 % of the token.  A replayed hipGraph pays 1.65 - 1.83 us per kernel node on this chip; part
// of that is the pair of agent-scope fences the runtime puts into every packet header (tools/aql/microbench_aql.cpp, per node:
// barrier bit + no fences 1.27 - 1.39 us, release only 1.37 - 1.49, acquire + release 1.53 - 1.80, system scope 3.0 - 3.5).  This queue
// writes the token's packets itself: header = {kernel dispatch, barrier bit, NO acquire, agent-scope release} -- the command
// processor still writes the L2s' dirty lines back when a launch ends, but no launch waits for its CU's caches to be invalidated:
// the kernels load every byte an earlier launch wrote past L1 instead (kernels.hip.h: the coherence rule).  The first launch of
// every token acquires at agent scope (it refreshes the scalar caches, through which {token, pos} are read); the first packet of
// a run acquires and the last one releases at system scope and carries the completion signal.
//
// Host-only C++ (no HIP): the kernels' code objects are read out of libllama2hip.so itself (the gfx950 entries of its offload
// bundles) and loaded through HSA; a launch is recorded by the NAME of its kernel (hipKernelNameRefByPtr in the caller).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct AqlQueue;
struct AqlProgram;      // one token's launches at one step level, recorded once, replayed for every token of that level

enum { AQL_FENCE_NONE = 0, AQL_FENCE_AGENT = 1, AQL_FENCE_SYSTEM = 2 };

// The GPU is the HSA agent at PCI (domain, bus, device, function); `so_path` = the shared library that holds the kernels.
// Returns null and a reason in `err` when the queue cannot be had (the caller then stays with replayed hipGraphs).
AqlQueue* aql_create(int pci_domain, int pci_bus, int pci_device, int pci_function, const char* so_path, char* err, size_t errlen);
void aql_destroy(AqlQueue* q);

// Forget every recorded kernel argument (call when ALL programs have been freed and no run is in flight).
void aql_reset(AqlQueue* q);
AqlProgram* aql_program_new(AqlQueue* q);
void aql_program_free(AqlProgram* p);
int aql_program_launches(const AqlProgram* p);

// One launch: `kernel_name` as the code object names it (mangled), explicit arguments packed as the kernel-argument segment lays
// them out (each at its natural alignment); the hidden arguments (block counts, group sizes, grid dimensions, dynamic LDS) are
// filled in here.  0 or -1 (reason: aql_last_error).
// flags: AQL_LAUNCH_ACQUIRES -- this launch acquires at agent scope whatever the run's fence setting: its kernel is not under the
// coherence rule (kernels.hip.h) and reads earlier launches' bytes through its CU's caches (the sampler's kernels).
enum { AQL_LAUNCH_ACQUIRES = 1 };
int aql_record(AqlQueue* q, AqlProgram* p, const char* kernel_name, const unsigned grid_blocks[3], const unsigned block[3],
               unsigned lds_dynamic, const void* args, size_t arg_bytes, unsigned flags = 0);
// Kernel arguments of everything recorded so far -> device memory.  Call once after recording, before the first run.
int aql_upload(AqlQueue* q);

// Run `ntok` tokens: token t replays per_token[t].  Blocking.  fence = acquire scope + 4 * release scope of the fences BETWEEN
// launches (AQL_FENCE_*; the run's first acquire and last release are always system scope).  elapsed_us (optional): first doorbell -> completion.
// A run that makes no progress (the queue's read index stands still) for L2_QUEUE_WAIT_S seconds (environment, read at aql_create; default
// 300) is given up: -1, and the queue is DEAD from then on (its ring still holds the run's packets) -- so is a queue whose error callback
// fired.  The caller destroys it and goes back to HIP launches.  elapsed_us == nullptr: the host sleeps on the completion signal instead of
// spinning (an untimed 2048-token 7B run is ~10 s).
int aql_run(AqlQueue* q, int ntok, AqlProgram* const* per_token, int fence, double* elapsed_us);
int aql_queue_dead(const AqlQueue* q);
const char* aql_last_error(const AqlQueue* q);
// How many gfx950 code objects the offload bundles of `so_path` hold (file parsing only: no GPU, no HSA); -1: the file cannot be read.
int aql_count_code_objects(const char* so_path);
