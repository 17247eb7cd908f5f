// Debug tool: run the integral-cubic inflection/weights arithmetic on host and device, print both.
#include "../contrast_renderer_amd/csrc/fill.hpp"
#include <cstdio>
using namespace crh;
struct Dump { float pb[12], d[4], disc, roots[9], w[16]; float cv[72]; float sv[12]; float n; };
struct RecSink { Dump* o; int nc=0, ns=0; __host__ __device__ void curve(float2 v, const float* w){ float* p=o->cv+6*nc; p[0]=v.x;p[1]=v.y;p[2]=w[0];p[3]=w[1];p[4]=w[2];p[5]=w[3]; nc++; } __host__ __device__ void solid(float2 v){ o->sv[2*ns]=v.x; o->sv[2*ns+1]=v.y; ns++; } __host__ __device__ void hull(float2){} };
__host__ __device__ void compute(const float* f, Dump& o) {
    Pt cp[4] = {vec_to_point(f[0], f[1]), vec_to_point(f[2], f[3]), vec_to_point(f[4], f[5]), vec_to_point(f[6], f[7])};
    Pt pb[4]; cubic_power_basis(cp, pb);
    for (int i = 0; i < 4; ++i) { o.pb[3*i] = pb[i].w; o.pb[3*i+1] = pb[i].x; o.pb[3*i+2] = pb[i].y; }
    float d[4]; inflection_coefficients(pb, true, d);
    for (int i = 0; i < 4; ++i) o.d[i] = d[i];
    Root r[3]; o.disc = integral_inflection_points(d, true, r);
    for (int i = 0; i < 3; ++i) { o.roots[3*i] = r[i].re; o.roots[3*i+1] = r[i].im; o.roots[3*i+2] = r[i].den; }
    float w[4][4]; cubic_weights(o.disc, r, w);
    for (int i = 0; i < 16; ++i) o.w[i] = w[i/4][i%4];
    for (int i = 0; i < 72; ++i) o.cv[i] = 0; for (int i = 0; i < 12; ++i) o.sv[i] = 0;
    RecSink sink{&o}; uint32_t err = 0; cubic_fill(cp, true, sink, err); o.n = (float)(sink.nc * 100 + sink.ns);
}
__global__ void k(const float* f, Dump* o, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) compute(f + 8*i, o[i]); }
int main() {
    std::vector<float> in; float v;
    while (scanf("%f", &v) == 1) in.push_back(v);
    int n = in.size() / 8;
    float* df; Dump* dd; hipMalloc(&df, in.size()*4); hipMalloc(&dd, n*sizeof(Dump));
    hipMemcpy(df, in.data(), in.size()*4, hipMemcpyHostToDevice);
    k<<<(n+63)/64, 64>>>(df, dd, n);
    std::vector<Dump> dev(n); hipMemcpy(dev.data(), dd, n*sizeof(Dump), hipMemcpyDeviceToHost);
    int shown = 0;
    for (int i = 0; i < n; ++i) {
        Dump h; compute(&in[8*i], h);
        const uint32_t* a = (const uint32_t*)&h; const uint32_t* b = (const uint32_t*)&dev[i];
        const char* names[] = {"pb","d","disc","roots","w","curve","solid","n"}; int begin[] = {0,12,16,17,26,42,114,126,127};
        for (int g = 0; g < 8; ++g) for (int k2 = begin[g]; k2 < begin[g+1]; ++k2) if (a[k2] != b[k2]) {
            if (shown++ < 40) printf("seg %d %s[%d]: host %.9g (%08x) dev %.9g (%08x)\n", i, names[g], k2-begin[g], ((float*)&h)[k2], a[k2], ((float*)&dev[i])[k2], b[k2]);
        }
    }
    for (int i = 0; i < n && i < 12; ++i) { Dump h; compute(&in[8*i], h); for (int v2 = 0; v2 < 6; ++v2) printf("seg %d v%d host %.9g %.9g %.9g %.9g %.9g | dev %.9g %.9g %.9g\n", i, v2, h.cv[6*v2], h.cv[6*v2+1], h.cv[6*v2+2], h.cv[6*v2+3], h.cv[6*v2+4], dev[i].cv[6*v2+2], dev[i].cv[6*v2+3], dev[i].cv[6*v2+4]); }
    printf("%d segments checked, %d differing words shown\n", n, shown);
}
