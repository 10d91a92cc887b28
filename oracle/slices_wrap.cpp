// TEST INFRASTRUCTURE (oracle/): the translation unit around two FREE FUNCTIONS of the reference that live in files which
// otherwise need TensorFlow headers:
//   * threenn_cpu   tf_ops/3d_interpolation/tf_interpolate.cpp:60-103   (the 3-NN search the ThreeNN op runs, CPU only)
//   * nnsearch      tf_ops/nn_distance/tf_nndistance.cpp:21-43          (the CPU twin of NmDistanceKernel's forward)
// Both are plain C loops with no dependency on anything else in their files.  oracle/Makefile (target `slices`) cuts exactly those
// line ranges out of /root/reference AT BUILD TIME into a scratch directory under /tmp, checks that the first and last line of each
// cut are the function's signature and its closing brace, compiles THIS file with `g++ -std=c++11 -O2 -fPIC -shared` (the flags of
// tf_interpolate_compile.sh:5 / tf_nndistance_compile.sh:8; no -march, no -ffast-math) with the two cuts #included below, and
// deletes the scratch directory.  No reference text is stored in the repository or travels anywhere; only the compiled
// oracle/_ref/libslices_ref.so does (like libinterp_ref.so).  No header, library or tool is stood in for: the cuts use nothing
// but the C language.
//
// Nothing here is an implementation of anything: this file contributes the two #include lines and one forwarding wrapper
// (nnsearch is `static` in the reference, so it needs an exported caller).
#ifndef SLICE_THREENN
#error "build through oracle/Makefile: make -C oracle slices"
#endif

extern "C" {
#include SLICE_THREENN   // void threenn_cpu(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx)
}

#include SLICE_NNSEARCH  // static void nnsearch(int b,int n,int m,const float * xyz1,const float * xyz2,float * dist,int * idx)

extern "C" void nnsearch_ref(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx) {
    nnsearch(b, n, m, xyz1, xyz2, dist, idx);
}
