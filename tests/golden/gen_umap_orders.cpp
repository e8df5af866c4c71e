#include <unordered_map>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
// prints iteration order after sequences of ops; format: one line per scenario
int main(){
  std::mt19937 rng(7);
  for(int sc=0; sc<400; ++sc){
    std::unordered_map<uint8_t,uint16_t> m;
    int reserve = sc%3==0 ? (int)(rng()%8) : -1;
    if(reserve>=0) m.reserve(reserve);
    int nops = 1 + rng()%30; int keyrange = (sc%2)? 13 : 40;
    printf("R %d", reserve);
    for(int i=0;i<nops;i++){
      int k = rng()%keyrange; int op = rng()%4;
      if(op==0){ m.erase((uint8_t)k); printf(" E%d",k);} else { m[(uint8_t)k]=1; printf(" I%d",k);} }
    printf(" =");
    for(auto&kv:m) printf(" %d",(int)kv.first);
    printf(" | nb %zu\n", m.bucket_count());
  }
}
