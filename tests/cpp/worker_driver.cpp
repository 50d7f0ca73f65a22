// worker_driver.cpp -- the reference GUI's threading pattern without Qt (qt_gui/src/mainwindow.cpp:46-68, 234-258):
// DirichletSolver::solve() runs on a WORKER thread, the iteration callback fires on that thread, and the "GUI" (main)
// thread calls requestStop() at an arbitrary moment.  Prints one JSON object for tests/test_gpu_cpp_compat.py.
//   worker_driver N poll_interval
#include "dirichlet_solver.hpp"

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <thread>

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    const int poll = argc > 2 ? std::atoi(argv[2]) : 0;
    Kokkos::initialize();
    DirichletSolver solver(N, N, 1.0, 2.0, 1.0, 2.0);
    solver.setVerbose(false);
    solver.setPollInterval(poll);
    // the GUI with every criterion unchecked: eps 0.0 and INT_MAX iterations (mainwindow.cpp:299-304) -- only a stop request ends it
    solver.setSolverParameters(0.0, 0.0, 0.0, INT_MAX);
    std::atomic<int> callbacks{0}, last_it{-1}, wrong_thread{0};
    std::atomic<bool> started{false};
    std::thread::id worker_id;
    solver.setIterationCallback([&](int it, double, double, double) {
        if (std::this_thread::get_id() != worker_id) wrong_thread = 1;        // invoked on the solving thread, synchronously
        ++callbacks; last_it = it;
        if (it >= 1) started = true;
    });
    SolverResults res;
    std::atomic<bool> done{false};
    std::thread worker([&] { worker_id = std::this_thread::get_id(); res = solver.solve(); done = true; });
    const auto t0 = std::chrono::steady_clock::now();
    while (!started && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(60)) std::this_thread::sleep_for(std::chrono::microseconds(200));
    std::this_thread::sleep_for(std::chrono::milliseconds(20));               // let it run: the request arrives mid-solve
    const bool was_running = !done;
    const auto t1 = std::chrono::steady_clock::now();
    solver.requestStop();                                                    // from the "GUI" thread
    worker.join();
    const double stop_latency_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    std::printf("{\"started\": %d, \"was_running\": %d, \"iterations\": %d, \"converged\": %d, \"stop_reason\": \"%s\", \"callbacks\": %d, "
                "\"last_callback_it\": %d, \"callback_on_worker_thread\": %d, \"main_is_not_worker\": %d, \"stop_latency_ms\": %.3f}\n",
                started ? 1 : 0, was_running ? 1 : 0, res.iterations, res.converged ? 1 : 0, res.stop_reason.c_str(), callbacks.load(),
                last_it.load(), wrong_thread ? 0 : 1, worker_id != std::this_thread::get_id() ? 1 : 0, stop_latency_ms);
    Kokkos::finalize();
    return 0;
}
