// TEST INFRASTRUCTURE ONLY: prints sizeof(CityFlow::Vehicle) of the reference as compiled by oracle/Makefile
// (the block size oracle/monotonic_new.cpp serves from its never-reusing arena).
#include <cstdio>

#include "vehicle/vehicle.h"

int main() {
    printf("%zu\n", sizeof(CityFlow::Vehicle));
    return 0;
}
