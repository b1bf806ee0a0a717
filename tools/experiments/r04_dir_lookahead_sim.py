"""Round 4, costed and not built: rejection rounds drawn AHEAD of the bounce that needs them.

random_on_hemisphere()'s rejection loop (raytracer.c:231-253) does not depend on the hit -- only the final flip does -- and
its place in a sample's RNG stream is known one draw (the roulette's) after the trip starts.  A lane that has its direction
after round 1 could therefore spend the trip's remaining rounds on the NEXT bounce's direction, speculatively, so that every
round runs at full occupancy and fewer rounds per trip would do (today: 3.84 rounds x 44 instructions per trip at 32 % of
the lanes).  This lane model (a path continues with p = 0.8, a continuing hit is diffuse with p = 0.9, a lane without a
direction at the end of a trip idles through the next one) gives, per trip, (share of lanes idling, attempts per lane):
    R = 4 rounds, no look-ahead (the shipped kernel)   4.6 %   (the kernels measure ~5 %)
    R = 3, no look-ahead                               9.8 %
    R = 3, one direction ahead                         3.5 %
    R = 2, one direction ahead                        11.7 %
so the best case saves one round of four (44 of ~850 instructions per trip, 5 %) against ~15 instructions of buffer
management, ten more VGPRs (the headline kernel has 86 of the 96 that five waves allow) and a saved RNG state per swap:
about 3 % at best, at the price of the fifth wave if the registers do not fit.  Not built."""
import random
p=0.5236
def sim(R,cap,q_cont=0.8,q_diff=0.9,trips=200000,spec=True):
    # lane: b = ready directions. each trip: if lane starved (needs dir, none) it idles this trip while doing rounds.
    b=0; need=False; starved_trips=0; useful=0; attempts_useful=0; att=0
    for t in range(trips):
        # rounds phase at start of trip? order: scan -> shade -> needs dir. model: at shade time need a dir.
        # production during this trip's rounds (after shade). simplify: rounds happen each trip; lane attempts while b<cap_eff
        if need:
            # waiting for direction from previous trip (carry-over): idle this trip's scan
            starved_trips+=1
            hit_now=False
        else:
            hit_now=True
        if hit_now:
            useful+=1
            r=random.random()
            if r<q_cont:
                if random.random()<q_diff: need=True
            else:
                b=0; need = random.random()<0.9*q_diff  # new path: camera ray hits diffuse likely ; buffer reset
        # rounds
        for k in range(R):
            capeff = cap if spec else (1 if need else 0)
            if need and b>0:
                b-=1; need=False
            if (need or b<capeff-0) and (spec or need):
                att+=1
                if random.random()<p:
                    if need: need=False
                    else: b+=1
        if need and b>0: b-=1; need=False
    return starved_trips/trips, att/trips
for R,cap,spec in [(4,0,False),(3,0,False),(2,0,False),(2,1,True),(2,2,True),(3,1,True),(3,2,True)]:
    print(R,cap,spec,sim(R,cap,spec=spec))
