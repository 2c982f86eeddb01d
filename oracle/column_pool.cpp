// TEST INFRASTRUCTURE -- part of the CPU oracle (see rt_oracle.h, column_pool.h).
#include "column_pool.h"

namespace orc
{
// ThreadPool.cpp:4-10
ColumnPool::ColumnPool(uint32_t numWorkers)
{
  workers.reserve(numWorkers);
  for (uint32_t i = 0; i < numWorkers; ++i)
    workers.emplace_back(&ColumnPool::workLoop, this);
}

// ThreadPool.cpp:12-23
ColumnPool::~ColumnPool()
{
  alive.store(false);
  {
    std::unique_lock guard(queueLock);
    queueCV.notify_all();
  }
  for (auto& t : workers)
    t.join();
}

// ThreadPool.cpp:40-55: publish the batch, then sleep until the queue is empty again
void ColumnPool::runBatch(JobBatch&& batch)
{
  {
    std::unique_lock guard(queueLock);
    numRemaining = (uint32_t)batch.jobs.size();
    pending.push_back(std::move(batch));
    queueCV.notify_all();
  }
  for (;;)
  {
    std::unique_lock guard(queueLock);
    if (pending.empty())
      return;
    completedCV.wait(guard);
  }
}

// ThreadPool.cpp:57-85: one job per lock acquisition, resumed outside the lock
void ColumnPool::workLoop()
{
  while (alive)
  {
    ColumnJob job;
    {
      std::unique_lock guard(queueLock);
      if (pending.empty() || pending.front().jobs.empty())
      {
        // (the reference waits unconditionally here; re-checking `alive` first only avoids a lost
        // wake-up at shutdown and does not change steady-state behaviour)
        if (!alive)
          break;
        queueCV.wait(guard);
        continue;
      }
      job = pending.front().jobs.front();
      pending.front().jobs.pop_front();
      numRunning++;
    }
    job.handle();
    {
      std::unique_lock guard(queueLock);
      numRemaining--;
      numRunning--;
      if (numRemaining == 0)
      {
        pending.pop_front();
        completedCV.notify_all();
      }
    }
  }
}
} // namespace orc
