// boost/timer/timer.hpp stand-in (test infrastructure): wall/user/system stopwatch with the cpu_timer interface the
// reference's TimeTracker uses (L/blt_util/time_util.hh).  Timing values never influence results.
#pragma once
#include <chrono>
#include <cstdint>
#include <ctime>
namespace boost
{
namespace timer
{
typedef std::int_least64_t nanosecond_type;
struct cpu_times
{
    nanosecond_type wall = 0, user = 0, system = 0;
    void clear() { wall = user = system = 0; }
};
class cpu_timer
{
public:
    cpu_timer() { start(); }
    bool is_stopped() const { return _stopped; }
    cpu_times elapsed() const
    {
        if (_stopped) return _t;
        cpu_times t(_t);
        t.wall += now_wall() - _w0;
        t.user += now_cpu() - _c0;
        return t;
    }
    void start()
    {
        _t.clear();
        _stopped = false;
        _w0 = now_wall();
        _c0 = now_cpu();
    }
    void stop()
    {
        if (_stopped) return;
        _t = elapsed();
        _stopped = true;
    }
    void resume()
    {
        if (!_stopped) return;
        _stopped = false;
        _w0 = now_wall();
        _c0 = now_cpu();
    }

private:
    static nanosecond_type now_wall()
    {
        return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    static nanosecond_type now_cpu() { return nanosecond_type(std::clock()) * (1000000000LL / CLOCKS_PER_SEC); }
    cpu_times _t;
    nanosecond_type _w0 = 0, _c0 = 0;
    bool _stopped = true;
};
} // namespace timer
} // namespace boost
