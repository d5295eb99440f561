#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int8_t* A /*16x64 row-major*/, const int8_t* B /*64x16: B[k][col] stored as Bt[col][k]*/, int* D /*16x16 row-major*/, int mode) {
    int l = threadIdx.x; int r = l & 15, g = l >> 4;
    v4i a, b, c = {0,0,0,0};
    // assumed: lane l holds A[row r][k = 16g + j], j=0..15 ; B[k=16g+j][col r]
    const int* ap = (const int*)(A + r*64 + 16*g);
    const int* bp = (const int*)(B + r*64 + 16*g);
    a = (v4i){ap[0],ap[1],ap[2],ap[3]};
    b = (v4i){bp[0],bp[1],bp[2],bp[3]};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    // assumed C/D: col = lane&15, row = (lane>>4)*4 + j
    for (int j=0;j<4;j++) D[(g*4+j)*16 + r] = c[j];
}
int main(){
    std::vector<int8_t> A(16*64), Bt(16*64); std::vector<int> D(256), R(256);
    for(int i=0;i<16*64;i++){ A[i] = (int8_t)((i*7+3)%11 - 5); Bt[i] = (int8_t)((i*5+1)%13 - 6);}
    for(int r=0;r<16;r++) for(int c=0;c<16;c++){ int s=0; for(int k=0;k<64;k++) s += A[r*64+k]*Bt[c*64+k]; R[r*16+c]=s; }
    int8_t *dA,*dB; int* dD; hipMalloc(&dA,1024); hipMalloc(&dB,1024); hipMalloc(&dD,1024);
    hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice); hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice);
    probe<<<1,64>>>(dA,dB,dD,0); hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost);
    int bad=0; for(int i=0;i<256;i++) if(D[i]!=R[i]) bad++;
    printf("mfma_i32_16x16x64_i8 layout check: %d mismatches\n", bad);
    if(bad){ for(int i=0;i<8;i++) printf("D[%d]=%d R=%d\n",i,D[i],R[i]); }
    return bad!=0;
}
