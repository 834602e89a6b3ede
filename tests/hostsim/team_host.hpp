// TEST-ONLY: the host runtime behind sztm::Team (subzero_amd/csrc/team.hpp): the G lanes of a team run as G fibers
// of one thread that meet at every cross-lane primitive.  A meeting checks what the GPU cannot: that all lanes still
// running arrive from the same call site (team-uniform control flow), and that no lane reads a value of a lane that has
// returned.  Context switches are a dozen instructions of x86-64 (no signal masks, no system calls), so millions of
// pairs can be pushed through the emulation in a CPU test.  Not part of the product.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../subzero_amd/csrc/team.hpp"

namespace teamhost {

// void fiber_switch(void **save_sp, void *load_sp): saves the callee-saved registers on the current stack, stores the
// stack pointer in *save_sp, switches to load_sp and restores
extern "C" void teamhost_fiber_switch(void **save_sp, void *load_sp);
__asm__(
    ".text\n"
    ".globl teamhost_fiber_switch\n"
    ".type teamhost_fiber_switch,@function\n"
    "teamhost_fiber_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size teamhost_fiber_switch,.-teamhost_fiber_switch\n");

struct Runtime : sztm::HostTeamRt {
  enum { MAXG = 32, STACK = 256 * 1024 };
  int G = 0;
  std::vector<char> stacks;
  void *sp[MAXG];
  void *main_sp = nullptr;
  unsigned char dead[MAXG];
  unsigned char deadsnap[2][MAXG];   // who had returned when the meeting of that parity took place
  sztm::u64 slot[2][MAXG];
  int site[MAXG], meets[MAXG];
  int cur = -1;
  void (*body)(void *, int) = nullptr;
  void *arg = nullptr;

  static void entry_thunk();
  static Runtime *&active() { static thread_local Runtime *r = nullptr; return r; }

  const sztm::u64 *meet(int lane, int s, sztm::u64 payload, const unsigned char **d) override {
    const int par = meets[lane] & 1;
    slot[par][lane] = payload;
    site[lane] = s;
    ++meets[lane];
    teamhost_fiber_switch(&sp[lane], main_sp);   // back to the scheduler; resumes when every live lane has arrived
    *d = deadsnap[par];
    return slot[par];
  }

  // runs fn(arg, lane) for lane = 0 .. g-1 as fibers
  void run(int g, void (*fn)(void *, int), void *a) {
    G = g; body = fn; arg = a;
    if (stacks.size() < (size_t)g * STACK) stacks.resize((size_t)g * STACK);
    for (int l = 0; l < g; ++l) {
      dead[l] = 0; meets[l] = 0; site[l] = -1;
      // initial frame: six zeroed callee-saved registers, then the return address = entry_thunk; the stack must be
      // 16-byte aligned + 8 at the thunk's first instruction, as after a call
      uintptr_t top = ((uintptr_t)(stacks.data() + (size_t)(l + 1) * STACK)) & ~(uintptr_t)15;
      void **f = (void **)(top - 8);          // slot that plays the thunk's own return address (never used)
      *f = nullptr;
      --f; *f = (void *)&Runtime::entry_thunk; // popped by `ret` in fiber_switch
      for (int k = 0; k < 6; ++k) { --f; *f = nullptr; }
      sp[l] = (void *)f;
    }
    Runtime *prev = active();
    active() = this;
    for (;;) {
      int live = 0;
      for (int l = 0; l < g; ++l) {
        if (dead[l]) continue;
        cur = l;
        teamhost_fiber_switch(&main_sp, sp[l]);
        if (!dead[l]) ++live;
      }
      if (live == 0) break;
      // every lane still running stands at a meeting: the same one
      int s0 = -1, m0 = -1;
      for (int l = 0; l < g; ++l) {
        if (dead[l]) continue;
        if (s0 < 0) { s0 = site[l]; m0 = meets[l]; }
        else if (site[l] != s0 || meets[l] != m0) {
          fprintf(stderr, "teamhost: lanes diverge at a cross-lane primitive: lane %d at site %d (line %d), another at site %d (line %d)\n",
                  l, site[l], site[l] / 256, s0, s0 / 256);
          abort();
        }
      }
      memcpy(deadsnap[(m0 - 1) & 1], dead, sizeof(dead));
    }
    active() = prev;
  }
};

inline void Runtime::entry_thunk() {
  Runtime *r = active();
  const int l = r->cur;
  r->body(r->arg, l);
  r->dead[l] = 1;
  void *dummy;
  teamhost_fiber_switch(&dummy, r->main_sp);
  abort();
}

// a lane that returns while others of its team wait at a meeting is a bug of the kernel body (on the GPU: a shuffle
// from an inactive lane).  Checked by the caller where it matters: after run(), `partial_exit` tells whether lanes left
// at different meeting counts.
inline bool exited_together(const Runtime &r) {
  for (int l = 1; l < r.G; ++l) if (r.meets[l] != r.meets[0]) return false;
  return true;
}

} // namespace teamhost
