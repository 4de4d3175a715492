// sk_broker -- the per-GPU server of libstrelka_amd.so's broker mode (csrc/sk_rt.h): the one process that holds a device's GPU
// context while any number of caller processes ($STRELKA_AMD_BROKER=1) use it.  Started on demand by the first client; by hand:
//   sk_broker --device D [--socket NAME] [--idle-exit SECONDS]
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" int sk_broker_serve(int device, const char* socket_name, int idle_seconds);

int main(int argc, char** argv)
{
    int device = 0, idle = 20;
    const char* name = "";
    if (const char* e = std::getenv("STRELKA_AMD_BROKER_IDLE_S")) idle = std::atoi(e);
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--socket") && i + 1 < argc) name = argv[++i];
        else if (!std::strcmp(argv[i], "--idle-exit") && i + 1 < argc) idle = std::atoi(argv[++i]);
        else {
            std::fprintf(stderr, "usage: sk_broker --device D [--socket NAME] [--idle-exit SECONDS]\n");
            return 64;
        }
    }
    return sk_broker_serve(device, name, idle < 1 ? 1 : idle);
}
