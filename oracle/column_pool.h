// TEST INFRASTRUCTURE -- part of the CPU oracle (see rt_oracle.h).
//
// Restatement of the reference's task runtime (src/ThreadPool.h:9-45,
// src/ThreadPool.cpp:4-85): a batch is a list of C++20 coroutine handles that
// start suspended; workers take ONE handle at a time under a single mutex,
// resume it outside the lock, and the worker that finishes the last job retires
// the batch and wakes the caller, which blocks in runBatch().  This is the
// harness the CPU baseline is timed on (one job per image column,
// src/cpu/CPURenderer.cpp:150-203), kept as it is on purpose.
#pragma once
#include <atomic>
#include <condition_variable>
#include <coroutine>
#include <cstdint>
#include <list>
#include <mutex>
#include <thread>
#include <vector>

namespace orc
{
// ThreadPool.h:9-20
struct ColumnJob
{
  struct promise_type
  {
    ColumnJob get_return_object() { return {std::coroutine_handle<promise_type>::from_promise(*this)}; }
    std::suspend_always initial_suspend() noexcept { return {}; }
    std::suspend_never final_suspend() noexcept { return {}; }
    void return_void() {}
    void unhandled_exception() {}
  };
  std::coroutine_handle<promise_type> handle;
};

// ThreadPool.h:21-24
struct JobBatch
{
  std::list<ColumnJob> jobs;
};

class ColumnPool
{
public:
  // ThreadPool.h:29: default hardware_concurrency() - 2
  explicit ColumnPool(uint32_t numWorkers);
  ~ColumnPool();
  void runBatch(JobBatch&& batch);
  uint32_t size() const { return (uint32_t)workers.size(); }

private:
  void workLoop();
  std::atomic_bool alive{true};
  std::mutex queueLock;
  std::condition_variable queueCV;
  std::condition_variable completedCV;
  uint32_t numRemaining = 0;
  uint32_t numRunning = 0;
  std::list<JobBatch> pending;
  std::vector<std::thread> workers;
};
} // namespace orc
