// scheme_runner — drives the C++ host layer (include/rsreg/*.hpp) from the command line so
// that the Python GPU tests can compare it with the ctypes path and the CPU checker.
//   scheme_runner <incremental|icp_edge|ndt_edge|icp_pair|ndt_pair|chain> <out_prefix> <a.pcd> <b.pcd> [...]
// chain: ChainRegistrar (consecutive pairs, RSREG_CHAIN_IN_FLIGHT=<K> of them side by side, default 3; RSREG_CHAIN_ITERATIONS=<n>:
// n fixed iterations at a 5 cm gate in the device-resident loop -- the bench's parameters -- instead of the reference's); the .txt
// holds every pair's `converged iterations context` line and 4x4, then the composed poses; no .pcd.
// Writes <out_prefix>.pcd (merged / aligned cloud) and <out_prefix>.txt (4x4 transforms, row-major).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rsreg/schemes.hpp"

using namespace rsreg;

static void dump(FILE *f, const Matrix4f &T)
{
    for (int r = 0; r < 4; ++r) std::fprintf(f, "%.9g %.9g %.9g %.9g\n", T(r, 0), T(r, 1), T(r, 2), T(r, 3));
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <mode> <out_prefix> <pcd> <pcd> [...]\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1], prefix = argv[2];
    const auto t_main = std::chrono::steady_clock::now();
    auto since_main = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_main).count(); };
    try {
        // RSREG_SCHEME_COLD=1: what the first registration() of a process is made of (main.cpp:85 calls it once per process).  The
        // steps a cold registration() goes through by itself are taken one by one first, each timed: the HIP runtime's start
        // (the first HIP call of the process), the context (streams, events, worker threads, first allocations); run 0 below
        // then shows what is left: code objects loaded at the first launch of every translation unit's kernels, buffers grown.
        const bool cold = std::getenv("RSREG_SCHEME_COLD") && std::getenv("RSREG_SCHEME_COLD")[0] == '1';
        if (cold) {
            const double a = since_main();
            int ndev = 0;
            (void)rsreg_device_count(&ndev);
            const double b = since_main();
            (void)Context::Default();
            const double c = since_main();
            std::fprintf(stderr, "%s cold: main() reached the runner + %.2f ms | first HIP call (runtime start, %d device%s) %.2f ms | context created %.2f ms\n",
                         mode.c_str(), a, ndev, ndev == 1 ? "" : "s", b - a, c - b);
        }
        std::vector<rgb_point_cloud_pointer> clouds;
        for (int i = 3; i < argc; ++i) {
            auto c = std::make_shared<rgb_point_cloud>();
            if (io::loadPCDFile(argv[i], *c) != 0) {
                std::fprintf(stderr, "cannot read %s\n", argv[i]);
                return 3;
            }
            clouds.push_back(c);
        }
        // RSREG_SCHEME_TIME=<reps>: the scheme is run <reps> times on fresh copies of the frames first and the wall time of
        // every run goes to stderr (frames on the host in, merged cloud on the host out; the first run also pays for
        // the context and the first allocations)
        const int timed_reps = std::getenv("RSREG_SCHEME_TIME") ? std::atoi(std::getenv("RSREG_SCHEME_TIME")) : 0;
        // RSREG_SCHEME_NO_STREAM=1: the merged cloud is built on the GPU and downloaded once at the end instead of being
        // streamed to the host frame by frame (schemes.hpp `stream_result`)
        const bool stream = !(std::getenv("RSREG_SCHEME_NO_STREAM") && std::getenv("RSREG_SCHEME_NO_STREAM")[0] == '1');
        const size_t chain_in_flight = std::getenv("RSREG_CHAIN_IN_FLIGHT") ? (size_t)std::atoi(std::getenv("RSREG_CHAIN_IN_FLIGHT")) : 3;
        const int chain_iterations = std::getenv("RSREG_CHAIN_ITERATIONS") ? std::atoi(std::getenv("RSREG_CHAIN_ITERATIONS")) : 0;
        auto chain_parameters = [&](ChainRegistrar &c) {
            if (chain_iterations <= 0) return;   // (the reference's: incremental_icp.hpp:46-49)
            c.params.max_iterations = chain_iterations;
            c.params.criteria_mode = RSREG_CRITERIA_FIXED;
            c.params.pipeline_mode = RSREG_PIPELINE_DEVICE_LOOP;
            c.params.max_correspondence_distance = 0.05;
        };
        std::unique_ptr<ChainRegistrar> chain;
        if (cold) std::fprintf(stderr, "%s cold: frames read from disk, main() + %.2f ms\n", mode.c_str(), since_main());
        for (int rep = 0; rep < timed_reps; ++rep) {
            std::vector<rgb_point_cloud_pointer> fresh;
            for (auto &c : clouds) fresh.push_back(std::make_shared<rgb_point_cloud>(*c));
            const bool host_loop_t = std::getenv("RSREG_SCHEME_HOST_LOOP") && std::getenv("RSREG_SCHEME_HOST_LOOP")[0] == '1';
            const auto t0 = std::chrono::steady_clock::now();
            size_t merged = 0;
            std::vector<double> frame_clock, stages;
            double finish_ms[4] = {0, 0, 0, 0};
            // the clock stops when registration() has returned the merged cloud; letting go of that cloud (157 MB) and of the
            // scheme object afterwards is the caller's business and is reported beside it
            rgb_point_cloud_pointer kept;
            double ms = 0;
            auto stop = [&] { ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
            if (mode == "incremental") {
                IncrementalICP s;
                s.device_resident = !host_loop_t;
                s.stream_result = stream;
                kept = s.registration(fresh);
                stop();
                frame_clock = s.frame_clock_ms;
                for (int q = 0; q < 4; ++q) finish_ms[q] = s.stream_finish_ms[q];
                stages.assign(s.stage_ms, s.stage_ms + 7);
            } else if (mode == "icp_edge") {
                ICPEdgeBasedRegistration s(-0.0261799f);
                s.device_resident = !host_loop_t;
                s.stream_result = stream;
                kept = s.registration(fresh);
                stop();
                frame_clock = s.frame_clock_ms;
                for (int q = 0; q < 4; ++q) finish_ms[q] = s.stream_finish_ms[q];
                stages.assign(s.stage_ms, s.stage_ms + 8);
            } else if (mode == "ndt_edge") {
                NDTEdgeBasedRegistration s(-0.0261799f);
                s.device_resident = !host_loop_t;
                s.stream_result = stream;
                kept = s.registration(fresh);
                stop();
                frame_clock = s.frame_clock_ms;
                for (int q = 0; q < 4; ++q) finish_ms[q] = s.stream_finish_ms[q];
                stages.assign(s.stage_ms, s.stage_ms + 8);
            } else if (mode == "chain") {
                // (the registrar lives across the repetitions, like a long-running caller's: run 0 pays for its contexts)
                if (!chain) chain.reset(new ChainRegistrar(chain_in_flight));
                chain_parameters(*chain);
                const auto poses = chain->registration(fresh);
                stop();
                std::fprintf(stderr, "chain run %d: %.3f ms per pair, %zu pairs, %zu in flight\n", rep, ms / std::max<size_t>(1, poses.size() - 1), poses.size() - 1,
                             chain->in_flight());
            }
            const double ms_scheme_gone = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            merged = kept ? kept->size() : 0;
            kept.reset();
            fresh.clear();
            const double ms_all_gone = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "%s run %d: %.2f ms, %zu frames, merged %zu points (then: scheme object destroyed + %.2f ms, clouds released + %.2f ms)\n",
                         mode.c_str(), rep, ms, clouds.size(), merged, ms_scheme_gone - ms, ms_all_gone - ms_scheme_gone);
            // RSREG_SCHEME_FRAMES=1: where the run's time went, frame by frame (schemes.hpp: frame_clock_ms)
            if (std::getenv("RSREG_SCHEME_FRAMES") && !frame_clock.empty()) {
                std::fprintf(stderr, "%s run %d frames: set-up %.2f |", mode.c_str(), rep, frame_clock[0]);
                for (size_t k = 1; k + 1 < frame_clock.size(); ++k) std::fprintf(stderr, " %.2f", frame_clock[k] - frame_clock[k - 1]);
                std::fprintf(stderr, " | merged cloud complete + %.2f (waited %.2f for frame 0's copy, %.2f for the downloads under way; handing the records over %.2f)\n", frame_clock.back() - frame_clock[frame_clock.size() - 2], finish_ms[0], finish_ms[1], finish_ms[3]);
                if (stages.size() == 8) {   // (the edge schemes: the caller's thread, call by call, over all frames -- schemes.hpp: stage_ms)
                    std::fprintf(stderr, "%s run %d calls: frames ahead queued %.2f | coarse alignment %.2f | refining ICP %.2f | transforms, grown target %.2f | result download queued %.2f ms  (coarse ICP: its source %.2f, the target's index %.2f, the alignment %.2f)\n",
                                 mode.c_str(), rep, stages[0], stages[1], stages[2], stages[3], stages[4], stages[5], stages[6], stages[7]);
                } else if (!stages.empty()) {   // (IncrementalICP: the caller's thread, call by call, over all frames -- schemes.hpp: stage_ms)
                    std::fprintf(stderr, "%s run %d calls: frames ahead queued %.2f | setInputSource %.2f | setInputTarget %.2f | align %.2f | transformPointCloud %.2f | += %.2f | result download queued %.2f ms\n",
                                 mode.c_str(), rep, stages[0], stages[1], stages[2], stages[3], stages[4], stages[5], stages[6]);
                }
            }
        }
        FILE *f = std::fopen((prefix + ".txt").c_str(), "w");
        rgb_point_cloud_pointer out;
        // RSREG_SCHEME_HOST_LOOP=1: every step of the frame loop on host clouds instead of cloud handles in HBM
        const bool host_loop = std::getenv("RSREG_SCHEME_HOST_LOOP") && std::getenv("RSREG_SCHEME_HOST_LOOP")[0] == '1';
        // RSREG_SCHEME_VERBOSE=1: the reference's progress lines on stdout; RSREG_SCHEME_BYPRODUCTS=<dir>: the edge-<k>.pcd /
        // edge_cloud.pcd files ICPEdgeBasedRegistration writes while it runs (icp_edge_based_registration.hpp:66-69,126)
        const bool verbose = std::getenv("RSREG_SCHEME_VERBOSE") && std::getenv("RSREG_SCHEME_VERBOSE")[0] == '1';
        const char *by_dir = std::getenv("RSREG_SCHEME_BYPRODUCTS");
        auto observables = [&](EdgeBasedRegistrationBase &s) {
            s.verbose = verbose;
            if (by_dir) { s.write_byproducts = true; s.byproduct_dir = by_dir; }
        };
        if (mode == "incremental") {
            IncrementalICP s;
            s.device_resident = !host_loop;
            s.stream_result = stream;
            out = s.registration(clouds);
            for (auto &T : s.transforms) dump(f, T);
        } else if (mode == "icp_edge" || mode == "ndt_edge") {
            const float rads = -0.0261799f;  // -1.5 deg per frame: the synthetic "bench" preset's yaw
            if (mode == "icp_edge") {
                ICPEdgeBasedRegistration s(rads);
                s.device_resident = !host_loop;
                s.stream_result = stream;
                observables(s);
                out = s.registration(clouds);
                for (auto &p : s.frame_transforms) { dump(f, p.first); dump(f, p.second); }
            } else {
                NDTEdgeBasedRegistration s(rads);
                s.device_resident = !host_loop;
                s.stream_result = stream;
                observables(s);
                out = s.registration(clouds);
                for (auto &p : s.frame_transforms) { dump(f, p.first); dump(f, p.second); }
            }
        } else if (mode == "chain") {
            if (!chain) chain.reset(new ChainRegistrar(chain_in_flight));
            chain_parameters(*chain);
            const auto poses = chain->registration(clouds);
            for (size_t k = 1; k < poses.size(); ++k) {
                std::fprintf(f, "%d %d %d\n", (int)chain->pair_results[k].converged, chain->pair_results[k].iterations, chain->pair_context[k]);
                dump(f, chain->pair_transforms[k]);
            }
            for (size_t k = 0; k < poses.size(); ++k) dump(f, poses[k]);
        } else if (mode == "icp_pair") {   // incremental_icp.hpp:57-63 on one pair, reference parameters
            IterativeClosestPoint<rgb_point, rgb_point> icp;
            detail::reference_icp_parameters(icp);
            icp.setInputSource(clouds[1]);
            icp.setInputTarget(clouds[0]);
            out = std::make_shared<rgb_point_cloud>();
            icp.align(*out);
            std::fprintf(f, "%d %d %d\n", (int)icp.hasConverged(), icp.result().iterations, icp.getConvergenceState());
            dump(f, icp.getFinalTransformation());
        } else if (mode == "ndt_pair") {
            NormalDistributionsTransform<rgb_point, rgb_point> ndt;
            ndt.setTransformationEpsilon(0.01);
            ndt.setStepSize(0.1);
            ndt.setResolution(1.0f);
            ndt.setMaximumIterations(50);
            ndt.setInputSource(clouds[1]);
            ndt.setInputTarget(clouds[0]);
            out = std::make_shared<rgb_point_cloud>();
            ndt.align(*out, Matrix4f::RotationY(0.01f));
            std::fprintf(f, "%d %d\n", (int)ndt.hasConverged(), ndt.getFinalNumIteration());
            dump(f, ndt.getFinalTransformation());
        } else {
            std::fprintf(stderr, "unknown mode %s\n", mode.c_str());
            return 2;
        }
        std::fclose(f);
        if (out) io::savePCDFileBinary(prefix + ".pcd", *out);
    } catch (const Error &e) {
        std::fprintf(stderr, "rsreg error %d: %s\n", e.status, e.what());
        return e.status == RSREG_ERR_NO_DEVICE ? 66 : 1;
    }
    return 0;
}
