// pose_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  Placeholder translation unit for the CPU
// restatement of the small dense pose solves; filled in when the
// GPU side of those rows lands.  Parity for these rows is UNPINNED: the reference delegates the
// arithmetic to Eigen::JacobiSVD / cv::SVD / cv::findEssentialMat, none of which is present.
extern "C" int oracle_pose_rows_available(void) { return 0; }
