// pcl_pin.cpp — records what PCL and the reference's own scheme classes compute on given inputs.
// TEST INFRASTRUCTURE (oracle/), built only by oracle/pcl_harness/CMakeLists.txt where PCL exists.
//
//   pcl_pin <input_dir> <output_dir>
// input_dir  : pair-0.pcd pair-1.pcd  (a target / source pair inside the reference's 1 cm gate)
//              chain-0.pcd .. chain-<n-1>.pcd (ORGANIZED frames for the scheme classes), guess.txt (4x4, row-major),
//              rads.txt (the per-frame yaw the edge schemes are constructed with)
// output_dir : one text file per result ("rows cols" then the values) and the merged clouds as .pcd
//
// The calls and constants below are the reference's: src/incremental_icp.hpp:36-66,
// src/ndt_edge_based_registration.hpp:32-112, src/icp_edge_based_registration.hpp:26-130, src/edge_extractor.hpp:7-39
// (the scheme classes and extract_edge_features are used unchanged through their public entry points,
// src/types.hpp:19,30-43).  Round 3 added sections (6)-(10): everything the engine grew in round 2.
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include <sys/stat.h>

#include <pcl/common/transforms.h>
#include <pcl/features/organized_edge_detection.h>   // (src/main.cpp:20 includes it ahead of edge_extractor.hpp)
#include <pcl/filters/approximate_voxel_grid.h>
#include <pcl/io/pcd_io.h>
#include <pcl/registration/correspondence_estimation.h>
#include <pcl/registration/correspondence_rejection_trimmed.h>
#include <pcl/registration/icp.h>
#include <pcl/registration/ndt.h>

// the reference's own headers, where they lie (-I ${REFERENCE_DIR}/src); utils.hpp holds float3
#include "utils.hpp"
#include "types.hpp"
#include "incremental_icp.hpp"
#include "edge_extractor.hpp"
#include "icp_edge_based_registration.hpp"
#include "ndt_edge_based_registration.hpp"

static void write_matrix(const std::string &path, const Eigen::MatrixXd &m)
{
    std::ofstream f(path);
    f << m.rows() << " " << m.cols() << "\n" << std::setprecision(17);
    for (int r = 0; r < m.rows(); ++r) {
        for (int c = 0; c < m.cols(); ++c) f << m(r, c) << (c + 1 < m.cols() ? " " : "\n");
    }
}

static rgb_point_cloud_pointer load(const std::string &path)
{
    rgb_point_cloud_pointer c(new rgb_point_cloud);
    if (pcl::io::loadPCDFile(path, *c) != 0) {
        std::cerr << "cannot read " << path << std::endl;
        std::exit(1);
    }
    return c;
}

int main(int argc, char **argv)
{
    if (argc != 3) {
        std::cerr << "usage: pcl_pin <input_dir> <output_dir>" << std::endl;
        return 2;
    }
    const std::string in = std::string(argv[1]) + "/", out = std::string(argv[2]) + "/";
    rgb_point_cloud_pointer tgt = load(in + "pair-0.pcd"), src = load(in + "pair-1.pcd");

    // ---- (1) first-iteration correspondences: CorrespondenceEstimation::determineCorrespondences,
    // the call ICP makes with max_dist = 0.01 (incremental_icp.hpp:47) -- pins the kd-tree's tie-break
    {
        pcl::registration::CorrespondenceEstimation<rgb_point, rgb_point> ce;
        ce.setInputSource(src);
        ce.setInputTarget(tgt);
        pcl::Correspondences corr;
        ce.determineCorrespondences(corr, 0.01);
        Eigen::MatrixXd m(corr.size(), 3);
        for (size_t i = 0; i < corr.size(); ++i) m.row(i) << corr[i].index_query, corr[i].index_match, (double)corr[i].distance;
        write_matrix(out + "corr_it0.txt", m);
    }
    // ---- (2) one pair through pcl::IterativeClosestPoint with the reference's constants
    {
        pcl::IterativeClosestPoint<rgb_point, rgb_point> icp;
        icp.setMaximumIterations(100);            // incremental_icp.hpp:46-49
        icp.setMaxCorrespondenceDistance(0.01);
        icp.setTransformationEpsilon(1);
        icp.setEuclideanFitnessEpsilon(1000);
        icp.setInputSource(src);
        icp.setInputTarget(tgt);
        rgb_point_cloud aligned;
        icp.align(aligned);
        write_matrix(out + "icp_reference_T.txt", icp.getFinalTransformation().cast<double>());
        Eigen::MatrixXd meta(1, 2);
        meta << (icp.hasConverged() ? 1 : 0), icp.getFitnessScore();
        write_matrix(out + "icp_reference_meta.txt", meta);
        // the same pair, fixed iteration counts without early exit (epsilons at PCL's defaults)
        for (int iters : {1, 5, 30}) {
            pcl::IterativeClosestPoint<rgb_point, rgb_point> it;
            it.setMaximumIterations(iters);
            it.setMaxCorrespondenceDistance(0.05);
            it.setInputSource(src);
            it.setInputTarget(tgt);
            rgb_point_cloud o;
            it.align(o);
            write_matrix(out + "icp_gate5cm_" + std::to_string(iters) + "it_T.txt", it.getFinalTransformation().cast<double>());
        }
    }
    // ---- (3) ApproximateVoxelGrid at the reference's 1 cm leaf (icp_edge...hpp:47) and at PCL's default
    for (int which = 0; which < 2; ++which) {
        pcl::ApproximateVoxelGrid<rgb_point> avg;
        if (which == 0) avg.setLeafSize(0.01f, 0.01f, 0.01f);
        avg.setInputCloud(src);
        rgb_point_cloud f;
        avg.filter(f);
        pcl::io::savePCDFileBinary(out + (which == 0 ? "voxel_1cm.pcd" : "voxel_default.pcd"), f);
    }
    // ---- (4) NDT with the reference's constants (ndt_edge...hpp:38-43) from the given guess
    {
        std::ifstream gf(in + "guess.txt");
        Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) gf >> guess(r, c);
        pcl::NormalDistributionsTransform<rgb_point, rgb_point> ndt;
        ndt.setTransformationEpsilon(0.01);
        ndt.setStepSize(0.1);
        ndt.setResolution(1.0);
        ndt.setMaximumIterations(50);
        ndt.setInputSource(src);
        ndt.setInputTarget(tgt);
        rgb_point_cloud aligned;
        ndt.align(aligned, guess);
        write_matrix(out + "ndt_reference_T.txt", ndt.getFinalTransformation().cast<double>());
        Eigen::MatrixXd meta(1, 3);
        meta << (ndt.hasConverged() ? 1 : 0), ndt.getFinalNumIteration(), ndt.getTransformationProbability();
        write_matrix(out + "ndt_reference_meta.txt", meta);
    }
    // ---- (5) the reference's scheme classes, unchanged, through their public entry point
    {
        std::vector<rgb_point_cloud_pointer> chain;
        for (int k = 0;; ++k) {
            std::ifstream probe(in + "chain-" + std::to_string(k) + ".pcd");
            if (!probe) break;
            chain.push_back(load(in + "chain-" + std::to_string(k) + ".pcd"));
        }
        if (chain.size() >= 2) {
            std::vector<rgb_point_cloud_pointer> copy;
            for (auto &f : chain) copy.push_back(rgb_point_cloud_pointer(new rgb_point_cloud(*f)));
            IncrementalICP scheme;
            rgb_point_cloud_pointer merged = scheme.registration(copy);
            pcl::io::savePCDFileBinary(out + "incremental_icp_merged.pcd", *merged);
        }
    }
    // ---- (6) the edge extractor, unchanged: label_indices[4] (RGB-Canny) of an ORGANIZED frame, as the cloud
    // extract_edge_features returns (src/edge_extractor.hpp:7-39) -- the points identify the pixel indices
    std::vector<rgb_point_cloud_pointer> frames;
    for (int k = 0;; ++k) {
        std::ifstream probe(in + "chain-" + std::to_string(k) + ".pcd");
        if (!probe) break;
        frames.push_back(load(in + "chain-" + std::to_string(k) + ".pcd"));
    }
    if (!frames.empty() && frames[0]->height > 1) {
        rgb_point_cloud_pointer copy(new rgb_point_cloud(*frames[0]));
        rgb_point_cloud_pointer edges = extract_edge_features(copy);
        pcl::io::savePCDFileBinary(out + "edge_features_chain0.pcd", *edges);
    }
    // ---- (7) the two edge-based schemes through registration() (types.hpp:30-43): merged clouds, and what
    // ICPEdgeBasedRegistration writes into ./dataset while it runs (icp_edge_based_registration.hpp:66-69,126)
    if (frames.size() >= 2 && frames[0]->height > 1) {
        float rads = -0.0026f;
        {
            std::ifstream rf(in + "rads.txt");
            if (rf) rf >> rads;
        }
        ::mkdir("dataset", 0755);   // the reference writes relative to the working directory
        {
            std::vector<rgb_point_cloud_pointer> copy;
            for (auto &f : frames) copy.push_back(rgb_point_cloud_pointer(new rgb_point_cloud(*f)));
            ICPEdgeBasedRegistration scheme(rads);
            rgb_point_cloud_pointer merged = scheme.registration(copy);
            pcl::io::savePCDFileBinary(out + "icp_edge_merged.pcd", *merged);
            for (size_t k = 0; k < frames.size(); ++k) {
                rgb_point_cloud e;
                if (pcl::io::loadPCDFile("dataset/edge-" + std::to_string(k) + ".pcd", e) == 0)
                    pcl::io::savePCDFileBinary(out + "icp_edge_byproduct_edge" + std::to_string(k) + ".pcd", e);
            }
            rgb_point_cloud grown;
            if (pcl::io::loadPCDFile("dataset/edge_cloud.pcd", grown) == 0) pcl::io::savePCDFileBinary(out + "icp_edge_byproduct_edge_cloud.pcd", grown);
        }
        {
            std::vector<rgb_point_cloud_pointer> copy;
            for (auto &f : frames) copy.push_back(rgb_point_cloud_pointer(new rgb_point_cloud(*f)));
            NDTEdgeBasedRegistration scheme(rads);
            rgb_point_cloud_pointer merged = scheme.registration(copy);
            pcl::io::savePCDFileBinary(out + "ndt_edge_merged.pcd", *merged);
        }
    }
    // ---- (8) reciprocal correspondences (Registration::setUseReciprocalCorrespondences; the engine's
    // rsreg_icp_params.use_reciprocal_correspondences): the first iteration's list and the reference-parameter transform
    {
        pcl::registration::CorrespondenceEstimation<rgb_point, rgb_point> ce;
        ce.setInputSource(src);
        ce.setInputTarget(tgt);
        pcl::Correspondences corr;
        ce.determineReciprocalCorrespondences(corr, 0.01);
        Eigen::MatrixXd m(corr.size(), 3);
        for (size_t i = 0; i < corr.size(); ++i) m.row(i) << corr[i].index_query, corr[i].index_match, (double)corr[i].distance;
        write_matrix(out + "corr_reciprocal_it0.txt", m);
        pcl::IterativeClosestPoint<rgb_point, rgb_point> icp;
        icp.setMaximumIterations(100);
        icp.setMaxCorrespondenceDistance(0.01);
        icp.setTransformationEpsilon(1);
        icp.setEuclideanFitnessEpsilon(1000);
        icp.setUseReciprocalCorrespondences(true);
        icp.setInputSource(src);
        icp.setInputTarget(tgt);
        rgb_point_cloud aligned;
        icp.align(aligned);
        write_matrix(out + "icp_reciprocal_T.txt", icp.getFinalTransformation().cast<double>());
    }
    // ---- (9) the trimmed rejector the reference constructs and never attaches (incremental_icp.hpp:38): attached here,
    // overlap ratio 0.8, the survivors of the first iteration's correspondences and the reference-parameter transform
    {
        pcl::registration::CorrespondenceEstimation<rgb_point, rgb_point> ce;
        ce.setInputSource(src);
        ce.setInputTarget(tgt);
        pcl::CorrespondencesPtr corr(new pcl::Correspondences);
        ce.determineCorrespondences(*corr, 0.01);
        pcl::registration::CorrespondenceRejectorTrimmed::Ptr trimmed(new pcl::registration::CorrespondenceRejectorTrimmed);
        trimmed->setOverlapRatio(0.8f);
        trimmed->setInputCorrespondences(corr);
        pcl::Correspondences kept;
        trimmed->getCorrespondences(kept);
        Eigen::MatrixXd m(kept.size(), 3);
        for (size_t i = 0; i < kept.size(); ++i) m.row(i) << kept[i].index_query, kept[i].index_match, (double)kept[i].distance;
        write_matrix(out + "corr_trimmed_it0.txt", m);
        pcl::IterativeClosestPoint<rgb_point, rgb_point> icp;
        icp.setMaximumIterations(100);
        icp.setMaxCorrespondenceDistance(0.01);
        icp.setTransformationEpsilon(1);
        icp.setEuclideanFitnessEpsilon(1000);
        pcl::registration::CorrespondenceRejectorTrimmed::Ptr attached(new pcl::registration::CorrespondenceRejectorTrimmed);
        attached->setOverlapRatio(0.8f);
        icp.addCorrespondenceRejector(attached);
        icp.setInputSource(src);
        icp.setInputTarget(tgt);
        rgb_point_cloud aligned;
        icp.align(aligned);
        write_matrix(out + "icp_trimmed_T.txt", icp.getFinalTransformation().cast<double>());
    }
    // ---- (10) a PCL-written binary_compressed file of the source frame (what a capture stored with
    // savePCDFileBinaryCompressed looks like to loadPCDFile, src/main.cpp:81): pins the LZF reader
    pcl::io::savePCDFileBinaryCompressed(out + "pair1_binary_compressed.pcd", *src);
    std::cout << "pcl_pin: wrote results to " << out << std::endl;
    return 0;
}
